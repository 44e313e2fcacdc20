// lasgun_amd/csrc/host.cpp -- scene description, OBJ reader, HLBVH builder, flattening.
// See host.h for the reference files each part mirrors.  Compiled with -ffp-contract=off.
#include "host.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <mutex>
#include <thread>

namespace lg {

// ------------------------------------------------------------------------------------------
// Materials (material/mod.rs:15-46, matte.rs:14-16)
// ------------------------------------------------------------------------------------------
Material material_matte(const double kd[3], double sigma) {
    Material m{};
    m.kind = MAT_MATTE;
    m.p[0] = kd[0]; m.p[1] = kd[1]; m.p[2] = kd[2];
    m.p[3] = fmin(fmax(sigma, 0.0), 90.0);
    return m;
}
Material material_default() {
    const double kd[3] = {0.5, 0.5, 0.5};
    return material_matte(kd, 0.0);
}

// ------------------------------------------------------------------------------------------
// Transform (space/transform.rs:49-197) on top of cgmath's column-combination products
// ------------------------------------------------------------------------------------------
static Mat4 mat_identity() {
    Mat4 r{};
    for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0;
    return r;
}
// lhs * rhs: result column j = ((a*r[j][0] + b*r[j][1]) + c*r[j][2]) + d*r[j][3], a..d = lhs columns
static Mat4 mat_mul(const Mat4 &l, const Mat4 &r) {
    Mat4 o;
    for (int j = 0; j < 4; ++j)
        for (int row = 0; row < 4; ++row)
            o.m[j][row] = ((l.m[0][row] * r.m[j][0] + l.m[1][row] * r.m[j][1]) + l.m[2][row] * r.m[j][2]) + l.m[3][row] * r.m[j][3];
    return o;
}
static Mat4 mat_transpose(const Mat4 &a) {
    Mat4 o;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) o.m[c][r] = a.m[r][c];
    return o;
}
Transform transform_identity() { return Transform{mat_identity(), mat_identity()}; }
void transform_concat_self(Transform &self, const Transform &other) { // transform.rs:191-197
    Mat4 m = mat_mul(other.m, self.m);
    Mat4 minv = mat_mul(self.minv, other.minv);
    self.m = m;
    self.minv = minv;
}
Transform transform_translate(const double d[3]) { // transform.rs:94-99
    Transform t = transform_identity();
    for (int i = 0; i < 3; ++i) { t.m.m[3][i] = d[i]; t.minv.m[3][i] = -d[i]; }
    return t;
}
Transform transform_scale(double x, double y, double z) { // transform.rs:101-108
    Transform t = transform_identity();
    t.m.m[0][0] = x; t.m.m[1][1] = y; t.m.m[2][2] = z;
    t.minv.m[0][0] = 1.0 / x; t.minv.m[1][1] = 1.0 / y; t.minv.m[2][2] = 1.0 / z;
    return t;
}
static inline double to_rad(double deg) { return deg * (PI / 180.0); } // cgmath Rad::from(Deg)
static Transform from_rotation(const Mat4 &m) { return Transform{m, mat_transpose(m)}; } // transform.rs:125-147
Transform transform_rotate_x(double deg) {
    double th = to_rad(deg), s, c;
    ::sincos(th, &s, &c); // Rad::sin_cos: glibc's sincos(), not sin() and cos() -- they differ in the last bit for some angles; g++ -O2 merged the pair anyway
    Mat4 m = mat_identity();
    m.m[1][1] = c; m.m[1][2] = s; m.m[2][1] = -s; m.m[2][2] = c;
    return from_rotation(m);
}
Transform transform_rotate_y(double deg) {
    double th = to_rad(deg), s, c;
    ::sincos(th, &s, &c); // Rad::sin_cos: glibc's sincos(), not sin() and cos() -- they differ in the last bit for some angles; g++ -O2 merged the pair anyway
    Mat4 m = mat_identity();
    m.m[0][0] = c; m.m[0][2] = -s; m.m[2][0] = s; m.m[2][2] = c;
    return from_rotation(m);
}
Transform transform_rotate_z(double deg) {
    double th = to_rad(deg), s, c;
    ::sincos(th, &s, &c); // Rad::sin_cos: glibc's sincos(), not sin() and cos() -- they differ in the last bit for some angles; g++ -O2 merged the pair anyway
    Mat4 m = mat_identity();
    m.m[0][0] = c; m.m[0][1] = s; m.m[1][0] = -s; m.m[1][1] = c;
    return from_rotation(m);
}
Transform transform_rotate(double deg, const double ax[3]) { // cgmath Matrix4::from_axis_angle
    double th = to_rad(deg), s, c;
    ::sincos(th, &s, &c);
    const double k = 1.0 - c;
    double x = ax[0], y = ax[1], z = ax[2];
    Mat4 m = mat_identity();
    m.m[0][0] = k * x * x + c;     m.m[0][1] = k * x * y + s * z; m.m[0][2] = k * x * z - s * y;
    m.m[1][0] = k * x * y - s * z; m.m[1][1] = k * y * y + c;     m.m[1][2] = k * y * z + s * x;
    m.m[2][0] = k * x * z + s * y; m.m[2][1] = k * y * z - s * x; m.m[2][2] = k * z * z + c;
    return from_rotation(m);
}

// ------------------------------------------------------------------------------------------
// Camera (camera.rs:61-102,155-194)
// ------------------------------------------------------------------------------------------
static double plane_height(bool persp, double param, double focal) {
    if (persp) return focal * std::tan(param * PI / 360.) * 2.;
    return param;
}
void Camera::init(bool persp, double p) {
    *this = Camera();
    perspective = persp;
    param = p;
    image_plane_height = plane_height(persp, p, 1.);
    pixel_separation = persp ? 0. : 1.;
}
void Camera::look_at(V3 o, V3 look, V3 upv) {
    V3 v = look - o;
    V3 a = cross(v, upv);
    origin = o;
    up = normalize(cross(a, v));
    aux = normalize(a);
    view = v;
    image_plane_height = plane_height(perspective, param, magnitude(v));
}
void Camera::set_supersampling(uint8_t base) {
    ss_root = (uint32_t)base + 1u;
    ss_distance = 1. / (double)ss_root;
}

// ------------------------------------------------------------------------------------------
// OBJ text (third-party `obj ^0.10` behaviour restated; triangle.rs:373-395)
// ------------------------------------------------------------------------------------------
namespace {
struct Cursor {
    const char *p, *end;
    bool eof() const { return p >= end; }
};
// next whitespace-separated word on the current line; false at end of line
bool next_word(Cursor &c, const char *&ws, const char *&we) {
    while (c.p < c.end && *c.p != '\n' && (*c.p == ' ' || *c.p == '\t' || *c.p == '\r' || *c.p == '\v' || *c.p == '\f')) ++c.p;
    if (c.p >= c.end || *c.p == '\n') return false;
    ws = c.p;
    while (c.p < c.end && *c.p != '\n' && !(*c.p == ' ' || *c.p == '\t' || *c.p == '\r' || *c.p == '\v' || *c.p == '\f')) ++c.p;
    we = c.p;
    return true;
}
void skip_line(Cursor &c) {
    while (c.p < c.end && *c.p != '\n') ++c.p;
    if (c.p < c.end) ++c.p;
}
float word_f32(const char *ws, const char *we, int line) {
    std::string w(ws, we);
    char *ep = nullptr;
    float v = std::strtof(w.c_str(), &ep); // correctly rounded, like Rust's str::parse::<f32>
    if (w.empty() || *ep != 0) throw Error("obj: bad number '" + w + "' on line " + std::to_string(line));
    return v;
}
long resolve_index(const char *s, const char *e, long count, int line) {
    std::string w(s, e);
    char *ep = nullptr;
    long v = std::strtol(w.c_str(), &ep, 10);
    if (w.empty() || *ep != 0) throw Error("obj: bad index '" + w + "' on line " + std::to_string(line));
    v = v < 0 ? count + v : v - 1;
    if (v < 0 || v >= count) throw Error("obj: index out of range on line " + std::to_string(line));
    return v;
}
} // namespace

void parse_obj_text(const char *text, size_t len, Obj &out) {
    Cursor c{text, text + len};
    int line = 0;
    while (!c.eof()) {
        ++line;
        const char *ws, *we;
        if (!next_word(c, ws, we)) { skip_line(c); continue; }
        std::string cmd(ws, we);
        if (cmd == "v" || cmd == "vn" || cmd == "vt") {
            int need = cmd == "vt" ? 2 : 3;
            float f[3] = {0, 0, 0};
            for (int i = 0; i < need; ++i) {
                if (!next_word(c, ws, we)) throw Error("obj: too few components on line " + std::to_string(line));
                f[i] = word_f32(ws, we, line);
            }
            std::vector<float> &dst = cmd == "v" ? out.position : (cmd == "vn" ? out.normal : out.texture);
            for (int i = 0; i < need; ++i) dst.push_back(f[i]);
        } else if (cmd == "f") {
            // Only the first three index tuples of a polygon are ever read (triangle.rs:41,47,53).
            for (int k = 0; k < 3; ++k) {
                if (!next_word(c, ws, we)) throw Error("obj: face with fewer than 3 vertices on line " + std::to_string(line));
                const char *s1 = ws;
                while (s1 < we && *s1 != '/') ++s1;
                Obj::Tuple tp{0, -1, -1};
                tp.v = (uint32_t)resolve_index(ws, s1, (long)out.position.size() / 3, line);
                if (s1 < we) {
                    const char *s2 = s1 + 1;
                    while (s2 < we && *s2 != '/') ++s2;
                    if (s2 > s1 + 1) tp.t = (int32_t)resolve_index(s1 + 1, s2, (long)out.texture.size() / 2, line);
                    if (s2 < we && we > s2 + 1) tp.n = (int32_t)resolve_index(s2 + 1, we, (long)out.normal.size() / 3, line);
                }
                out.tri.push_back(tp);
            }
        } else if (cmd == "o" || cmd == "g" || cmd == "s" || cmd == "mtllib" || cmd == "usemtl" || cmd[0] == '#') {
            // grouping never changes the order of `f` lines, which is TriangleIterator's order (triangle.rs:315-371)
        } else {
            throw Error("obj: unexpected command '" + cmd + "' on line " + std::to_string(line));
        }
        skip_line(c);
    }
}

// ------------------------------------------------------------------------------------------
// Bounds (space/bounds.rs) and their transform (space/transform.rs:219-240)
// ------------------------------------------------------------------------------------------
namespace {
Bounds b_new(V3 p0, V3 p1) {
    return Bounds{V3{bmin(p0.x, p1.x), bmin(p0.y, p1.y), bmin(p0.z, p1.z)}, V3{bmax(p0.x, p1.x), bmax(p0.y, p1.y), bmax(p0.z, p1.z)}};
}
Bounds b_none() { return Bounds{V3{F64_MAX, F64_MAX, F64_MAX}, V3{-F64_MAX, -F64_MAX, -F64_MAX}}; }
Bounds b_union(const Bounds &a, const Bounds &b) {
    return Bounds{V3{bmin(a.min.x, b.min.x), bmin(a.min.y, b.min.y), bmin(a.min.z, b.min.z)},
                  V3{bmax(a.max.x, b.max.x), bmax(a.max.y, b.max.y), bmax(a.max.z, b.max.z)}};
}
Bounds b_add_point(const Bounds &a, V3 p) {
    return Bounds{V3{bmin(a.min.x, p.x), bmin(a.min.y, p.y), bmin(a.min.z, p.z)}, V3{bmax(a.max.x, p.x), bmax(a.max.y, p.y), bmax(a.max.z, p.z)}};
}
double b_area(const Bounds &b) { // bounds.rs:110-114
    V3 d = b.max - b.min;
    double half = d.x * d.y + d.x * d.z + d.y * d.z;
    return half + half;
}
int b_max_extent(const Bounds &b) { // bounds.rs:125-130: `d.z > d.z` is never true, so never 0 (kept)
    V3 d = b.max - b.min;
    if (d.x > d.y && d.z > d.z) return 0;
    if (d.y > d.z) return 1;
    return 2;
}
V3 b_offset(const Bounds &b, V3 p) { // bounds.rs:133-139
    V3 o = p - b.min;
    if (b.max.x > b.min.x) o.x /= b.max.x - b.min.x;
    if (b.max.y > b.min.y) o.y /= b.max.y - b.min.y;
    if (b.max.z > b.min.z) o.z /= b.max.z - b.min.z;
    return o;
}
Bounds b_transform(const Mat4 &m, const Bounds &b) {
    V3 lo{0, 0, 0}, hi{0, 0, 0};
    double *plo = &lo.x, *phi = &hi.x;
    for (int row = 0; row < 3; ++row) {
        double xa = m.m[0][row] * b.min.x, xb = m.m[0][row] * b.max.x;
        double ya = m.m[1][row] * b.min.y, yb = m.m[1][row] * b.max.y;
        double za = m.m[2][row] * b.min.z, zb = m.m[2][row] * b.max.z;
        plo[row] = ((bmin(xa, xb) + bmin(ya, yb)) + bmin(za, zb)) + m.m[3][row];
        phi[row] = ((bmax(xa, xb) + bmax(ya, yb)) + bmax(za, zb)) + m.m[3][row];
    }
    return b_new(lo, hi);
}

// ------------------------------------------------------------------------------------------
// HLBVH build (bvh.rs:164-453,575-635), index-based
// ------------------------------------------------------------------------------------------
struct LinNode {
    Bounds b;
    bool leaf;
    uint32_t a, c; // leaf: (prim offset, nprims as u16) ; interior: (axis, second child)
};
struct BuiltBVH {
    std::vector<LinNode> nodes;
    std::vector<uint32_t> order;
};

uint32_t spread3(uint32_t x) { // bvh.rs:590-598
    if (x == 1024u) x = 1023u;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
uint32_t morton_zyz(V3 v) { // bvh.rs:575-579: z, y, z -- x never contributes (kept)
    return (spread3(as_u32(v.z)) << 2) | (spread3(as_u32(v.y)) << 1) | spread3(as_u32(v.z));
}

class Builder {
  public:
    Builder(const std::vector<Bounds> &prim_bounds, size_t max_prims) : pb_(prim_bounds) {
        leaf_limit_ = (uint32_t)(uint8_t)(max_prims < 255 ? max_prims : 255); // bvh.rs:187
    }
    BuiltBVH run() {
        size_t n = pb_.size();
        if (n == 0) throw Error("empty aggregate: the reference recurses without bound in build_upper_sah (bvh.rs:355-424)");
        const bool TT = n > 50000 && std::getenv("LASGUN_DEBUG_TIMES");
        auto tnow = [] { return std::chrono::steady_clock::now(); };
        auto tms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto q0 = tnow();
        out_.order.assign(n, 0xFFFFFFFFu);
        centroid_.resize(n);
        Bounds all = b_none();
        for (size_t i = 0; i < n; ++i) {
            centroid_[i] = 0.5 * pb_[i].min + 0.5 * pb_[i].max; // bvh.rs:530
            all = b_union(all, pb_[i]);
        }
        const auto q1 = tnow();
        // Morton codes + stable LSD radix sort, 5 passes of 6 bits (bvh.rs:217-229,600-635)
        std::vector<std::pair<uint32_t, uint32_t>> cur(n), tmp(n); // (code, prim)
        for (size_t i = 0; i < n; ++i) cur[i] = {morton_zyz(b_offset(all, centroid_[i]) * 1024.0), (uint32_t)i};
        for (int pass = 0; pass < 5; ++pass) {
            int shift = 6 * pass;
            size_t hist[64] = {0};
            for (auto &e : cur) hist[(e.first >> shift) & 63]++;
            size_t pos[64];
            size_t acc = 0;
            for (int b = 0; b < 64; ++b) { pos[b] = acc; acc += hist[b]; }
            for (auto &e : cur) tmp[pos[(e.first >> shift) & 63]++] = e;
            cur.swap(tmp);
        }
        code_.resize(n); prim_.resize(n);
        for (size_t i = 0; i < n; ++i) { code_[i] = cur[i].first; prim_[i] = cur[i].second; }
        const auto q2 = tnow();
        // treelets: runs of equal top-12 bits (bvh.rs:240-265)
        std::vector<int32_t> roots;
        size_t start = 0;
        for (size_t end = 1; end <= n; ++end) {
            if (end == n || ((code_[start] & 0x3FFC0000u) != (code_[end] & 0x3FFC0000u))) {
                roots.push_back(emit(start, end - start, 17));
                start = end;
            }
        }
        const auto q3 = tnow();
        int32_t root = upper(roots.data(), roots.size(), 0);
        const auto q4 = tnow();
        out_.nodes.resize(pool_.size() - dead_);
        uint32_t off = 0;
        emit_linear(root, off);
        out_.nodes.resize(off);
        if (TT) std::fprintf(stderr, "[lasgun] BVH of %zu: centroids %.3f, morton + sort %.3f, treelets %.3f (%zu roots), upper SAH %.3f, linearise %.3f ms\n", n, tms(q0, q1), tms(q1, q2), tms(q2, q3), roots.size(), tms(q3, q4), tms(q4, tnow()));
        return std::move(out_);
    }

  private:
    struct BNode {
        Bounds b;
        int32_t c0, c1; // c0 < 0: leaf
        uint32_t first, count;
        int axis;
    };
    int32_t leaf(uint32_t first, uint32_t count, const Bounds &b) {
        pool_.push_back(BNode{b, -1, -1, first, count, 0});
        return (int32_t)pool_.size() - 1;
    }
    int32_t interior(int axis, int32_t c0, int32_t c1) {
        pool_.push_back(BNode{b_union(pool_[c0].b, pool_[c1].b), c0, c1, 0, 0, axis});
        return (int32_t)pool_.size() - 1;
    }
    // bvh.rs:278-347 over the sorted slice [s, s+n)
    int32_t emit(size_t s, size_t n, int bit) {
        for (;;) {
            if (bit == -1 || n < (size_t)leaf_limit_) {
                uint32_t first = next_order_;
                Bounds b = b_none();
                for (size_t i = 0; i < n; ++i) {
                    out_.order[first + i] = prim_[s + i];
                    b = b_union(b, pb_[prim_[s + i]]);
                }
                next_order_ += (uint32_t)n;
                return leaf(first, (uint32_t)n, b);
            }
            uint32_t mask = 1u << bit;
            if ((code_[s] & mask) == (code_[s + n - 1] & mask)) { --bit; continue; }
            size_t lo = 0, hi = n - 1;
            while (lo + 1 != hi) {
                size_t mid = (lo + hi) / 2;
                if ((code_[s + lo] & mask) == (code_[s + mid] & mask)) lo = mid; else hi = mid;
            }
            int32_t c0 = emit(s, hi, bit - 1);
            int32_t c1 = emit(s + hi, n - hi, bit - 1);
            return interior(bit % 3, c0, c1);
        }
    }
    size_t bucket_of(int32_t node, int dim, const Bounds &cb) const { // bvh.rs:383-387,415-419
        const Bounds &b = pool_[node].b;
        double centroid = (comp(b.min, dim) + comp(b.max, dim)) * 0.5;
        double f = (centroid - comp(cb.min, dim)) / (comp(cb.max, dim) - comp(cb.min, dim));
        size_t k = (size_t)as_u32(12.0 * f);
        return k == 12 ? 11 : k;
    }
    // bvh.rs:350-427
    int32_t upper(int32_t *roots, size_t n, int depth) {
        if (n == 1) return roots[0];
        if (n == 0 || depth > 4096)
            throw Error("degenerate upper-SAH split: the reference recurses without bound here (bvh.rs:414-424)");
        Bounds all = b_none(), cb = b_none();
        for (size_t i = 0; i < n; ++i) {
            const Bounds &b = pool_[roots[i]].b;
            all = b_union(all, b);
            cb = b_add_point(cb, 0.5 * (b.min + b.max));
        }
        int dim = b_max_extent(cb);
        size_t count[12] = {0};
        Bounds bb[12];
        for (auto &x : bb) x = b_none();
        for (size_t i = 0; i < n; ++i) {
            size_t k = bucket_of(roots[i], dim, cb);
            if (k > 11) throw Error("SAH bucket index out of range (the reference would panic)");
            count[k]++;
            bb[k] = b_union(bb[k], pool_[roots[i]].b);
        }
        double cost[12];
        for (int i = 0; i < 12; ++i) {
            Bounds b0 = b_none(), b1 = b_none();
            size_t n0 = 0, n1 = 0;
            for (int j = 0; j <= i; ++j) { b0 = b_union(b0, bb[j]); n0 += count[j]; }
            for (int j = i + 1; j < 12; ++j) { b1 = b_union(b1, bb[j]); n1 += count[j]; }
            cost[i] = 0.125 + ((double)n0 * b_area(b0) + (double)n1 * b_area(b1)) / b_area(all);
        }
        size_t split = 0;
        for (size_t i = 0; i < 12; ++i)
            if (cost[i] < cost[split]) split = i;
        // `partition ^0.1`: in-place two-pointer swap partition (third-party; parity unpinned)
        size_t mid;
        {
            size_t l = 0, r = n - 1;
            for (;;) {
                while (l < n && bucket_of(roots[l], dim, cb) <= split) ++l;
                while (r > 0 && !(bucket_of(roots[r], dim, cb) <= split)) --r;
                if (l >= r) { mid = l; break; }
                std::swap(roots[l], roots[r]);
            }
        }
        int32_t lo = upper(roots, mid, depth + 1);
        int32_t hi = upper(roots + mid, n - mid, depth + 1);
        return interior(dim, lo, hi);
    }
    uint32_t emit_linear(int32_t node, uint32_t &off) { // bvh.rs:430-453
        uint32_t my = off++;
        const BNode &bn = pool_[node];
        out_.nodes[my].b = bn.b;
        if (bn.c0 < 0) {
            out_.nodes[my].leaf = true;
            out_.nodes[my].a = bn.first;
            out_.nodes[my].c = (uint32_t)(uint16_t)bn.count; // `nprims as u16` (bvh.rs:440)
        } else {
            emit_linear(bn.c0, off);
            uint32_t second = emit_linear(bn.c1, off);
            out_.nodes[my].leaf = false;
            out_.nodes[my].a = (uint32_t)(uint8_t)bn.axis;
            out_.nodes[my].c = second;
        }
        return my;
    }

    const std::vector<Bounds> &pb_;
    std::vector<V3> centroid_;
    std::vector<uint32_t> code_, prim_;
    std::vector<BNode> pool_;
    size_t dead_ = 0;
    uint32_t leaf_limit_ = 0;
    uint32_t next_order_ = 0;
    BuiltBVH out_;
};

// ------------------------------------------------------------------------------------------
// Fast tree (NOT in the reference): binned-SAH BVH with one primitive per leaf over the same
// primitive boxes, emitted in the same pre-order format as the reference tree so the device
// tables and node records are shared.  Used only by the opt-in fast traversal mode.
// ------------------------------------------------------------------------------------------
class FastBuilder {
  public:
    // leaf_max: primitives per leaf.  Measured under the wide walk (config 3 / 4 / 4m / 5, ms): 4 -> 8.17 / 18.9 / 10.4 / 36.9,
    // 2 -> 7.97 / 18.0 / 9.87 / 35.9, 1 -> 7.98 / 17.3 / 9.42 / 36.3 -- a primitive test costs more than a box test, and a
    // nested accel that shares a leaf with a wall is entered (ray transform, three divisions, a level frame) by every ray
    // that reaches the wall: with the mesh's tree at 1, config 4 takes 17.3 / 18.8 / 19.6 ms with the aggregates' at 1 / 2 / 4.
    FastBuilder(const std::vector<Bounds> &pb, size_t leaf_max) : pb_(pb), LEAF_MAX(leaf_max) {}
    BuiltBVH run() {
        size_t n = pb_.size();
        idx_.resize(n);
        for (size_t i = 0; i < n; ++i) idx_[i] = (uint32_t)i;
        cen_.resize(n);
        for (size_t i = 0; i < n; ++i) cen_[i] = 0.5 * (pb_[i].min + pb_[i].max);
        out_.order.reserve(n);
        build(0, n);
        return std::move(out_);
    }

  private:
    uint32_t build(size_t s, size_t e) {
        uint32_t my = (uint32_t)out_.nodes.size();
        out_.nodes.push_back(LinNode{b_none(), true, 0, 0});
        Bounds b = b_none(), cb = b_none();
        for (size_t i = s; i < e; ++i) { b = b_union(b, pb_[idx_[i]]); cb = b_add_point(cb, cen_[idx_[i]]); }
        size_t n = e - s;
        if (n <= LEAF_MAX) {
            out_.nodes[my] = LinNode{b, true, (uint32_t)out_.order.size(), (uint32_t)n};
            for (size_t i = s; i < e; ++i) out_.order.push_back(idx_[i]);
            return my;
        }
        // peel off primitives that are as large as the node itself (walls, nested groups): left in
        // place they would inflate every box of the subtree they end up in
        {
            double na = b_area(b);
            auto big = [&](uint32_t p) { return b_area(pb_[p]) >= 0.35 * na; };
            size_t nbig = 0;
            for (size_t i = s; i < e; ++i) nbig += big(idx_[i]) ? 1 : 0;
            if (nbig > 0 && nbig < n && na > 0.0 && std::isfinite(na)) {
                auto it = std::stable_partition(idx_.begin() + s, idx_.begin() + e, big);
                size_t m = (size_t)(it - idx_.begin());
                build(s, m);
                uint32_t second = build(m, e);
                out_.nodes[my] = LinNode{b, false, 0u, second};
                return my;
            }
        }
        V3 ext = cb.max - cb.min;
        int axis = ext.x >= ext.y ? (ext.x >= ext.z ? 0 : 2) : (ext.y >= ext.z ? 1 : 2);
        double lo = comp(cb.min, axis), width = comp(ext, axis);
        size_t mid = s + n / 2;
        bool split_found = false;
        if (width > 0.0 && std::isfinite(width)) {
            const int NB = 16;
            size_t cnt[NB] = {0};
            Bounds bb[NB];
            for (auto &x : bb) x = b_none();
            auto bin_of = [&](uint32_t p) {
                int k = (int)((comp(cen_[p], axis) - lo) / width * NB);
                return k < 0 ? 0 : (k >= NB ? NB - 1 : k);
            };
            for (size_t i = s; i < e; ++i) { int k = bin_of(idx_[i]); cnt[k]++; bb[k] = b_union(bb[k], pb_[idx_[i]]); }
            double best = INFINITY;
            int best_k = -1;
            for (int k = 0; k < NB - 1; ++k) {
                Bounds l = b_none(), r = b_none();
                size_t nl = 0, nr = 0;
                for (int j = 0; j <= k; ++j) { l = b_union(l, bb[j]); nl += cnt[j]; }
                for (int j = k + 1; j < NB; ++j) { r = b_union(r, bb[j]); nr += cnt[j]; }
                if (nl == 0 || nr == 0) continue;
                double c = (double)nl * b_area(l) + (double)nr * b_area(r);
                if (c < best) { best = c; best_k = k; }
            }
            if (best_k >= 0) {
                auto it = std::stable_partition(idx_.begin() + s, idx_.begin() + e, [&](uint32_t p) { return bin_of(p) <= best_k; });
                mid = (size_t)(it - idx_.begin());
                split_found = mid > s && mid < e;
            }
        }
        if (!split_found) mid = s + n / 2; // coincident centroids: split the list in half
        build(s, mid);
        uint32_t second = build(mid, e);
        out_.nodes[my] = LinNode{b, false, (uint32_t)axis, second};
        return my;
    }
    const std::vector<Bounds> &pb_;
    const size_t LEAF_MAX;
    std::vector<uint32_t> idx_;
    std::vector<V3> cen_;
    BuiltBVH out_;
};

// ------------------------------------------------------------------------------------------
// Flattening
// ------------------------------------------------------------------------------------------
struct MeshTables { // per scene mesh, shared by all its instances
    bool built = false;
    uint32_t node_base = 0, prim_base = 0, nnodes = 0, norder = 0;
    uint32_t max_stack = 0;
    uint32_t fnode_base = 0, fprim_base = 0, fmax_stack1 = 0;
    Bounds root_bounds{};
    double cmax = 0.0; // largest |coordinate| of its triangle boxes
    uint32_t tri_base = 0;
    bool has_n = false, has_uv = false;
    BuiltBVH bvh; // kept for the structure dump
};

struct Flattener {
    const Scene &scene;
    FlatScene &out;
    std::vector<MeshTables> meshes;
    bool with_fast = false;
    double mesh_build_ms = 0.0; // (LASGUN_DEBUG_TIMES)

    int32_t add_material(const Material &m) {
        DMaterial d{};
        d.kind = m.kind;
        std::memcpy(d.p, m.p, sizeof d.p);
        out.materials.push_back(d);
        if (m.kind == MAT_GLASS || m.kind == MAT_MIRROR) out.has_specular = true;
        return (int32_t)out.materials.size() - 1;
    }

    static Affine to_affine(const Mat4 &m) {
        Affine a;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 3; ++r) a.c[c][r] = m.m[c][r];
        // bottom row must be (0,0,0,1) for the 3x4 shortcut to be exact
        if (m.m[0][3] != 0.0 || m.m[1][3] != 0.0 || m.m[2][3] != 0.0 || m.m[3][3] != 1.0)
            throw Error("non-affine transform");
        return a;
    }

    // worst-case traversal stack use of one BVH; child_extra(prim ref) = extra entries when a leaf holds a child accel
    template <class F> static uint32_t stack_need(const std::vector<LinNode> &nodes, const std::vector<uint32_t> &refs_in_order, F child_extra) {
        uint32_t worst = 0;
        struct It { uint32_t node, depth; };
        std::vector<It> st{{0, 0}};
        while (!st.empty()) {
            It it = st.back(); st.pop_back();
            const LinNode &n = nodes[it.node];
            if (n.leaf) {
                uint32_t extra = 0;
                for (uint32_t i = 0; i < n.c; ++i) extra = std::max(extra, child_extra(refs_in_order[n.a + i]));
                worst = std::max(worst, it.depth + extra);
            } else {
                st.push_back({it.node + 1, it.depth + 1});
                st.push_back({n.c, it.depth + 1});
            }
        }
        return worst;
    }

    // Constants of the pruned walk for one BVHAccel level (DAccel::prune), from its root box, the
    // largest coordinate magnitude of its boxes and its smallest sphere radius (0 = the level holds no sphere).
    static void prune_constants(const Bounds &root, double cmax, double rmin, bool has_sphere, double out6[6]) {
        const V3 ext = root.max - root.min;
        const double size = ext.x + ext.y + ext.z;
        out6[0] = 0.5 * (root.min.x + root.max.x); out6[1] = 0.5 * (root.min.y + root.max.y); out6[2] = 0.5 * (root.min.z + root.max.z);
        out6[3] = size;
        bool ok = std::isfinite(size) && std::isfinite(cmax) && size >= 1.0 / PRUNE_RANGE && cmax <= PRUNE_RANGE &&
                  std::isfinite(out6[0]) && std::isfinite(out6[1]) && std::isfinite(out6[2]);
        if (has_sphere) ok = ok && std::isfinite(rmin) && rmin >= 1.0 / PRUNE_RANGE; // (a radius <= 0 or NaN never passes)
        out6[4] = ok ? PRUNE_E0_PER_COORD * cmax : INFINITY;
        out6[5] = has_sphere && ok ? PRUNE_E2_TIMES_RMIN / rmin : 0.0;
    }
    static double bounds_cmax(const std::vector<Bounds> &pb) {
        double m = 0.0;
        for (const Bounds &b : pb)
            for (int a = 0; a < 3; ++a) {
                const double lo = std::fabs(comp(b.min, a)), hi = std::fabs(comp(b.max, a));
                if (!(lo <= m)) m = lo; // (a NaN bound makes the maximum NaN: the level is then never pruned)
                if (!(hi <= m)) m = hi;
            }
        return m;
    }
    // per node of a reference tree: does a leaf below it hold a nested accel (NODE_NOPRUNE)?
    static std::vector<char> nodes_over_accels(const BuiltBVH &bvh, const std::vector<uint32_t> &ref) {
        std::vector<char> np(bvh.nodes.size(), 0);
        for (size_t i = bvh.nodes.size(); i-- > 0;) { // children follow their parent in the linear order
            const LinNode &n = bvh.nodes[i];
            if (n.leaf) { for (uint32_t k = 0; k < (n.c & 0xFFFFu); ++k) if ((ref[bvh.order[n.a + k]] >> 30) == PK_ACCEL) np[i] = 1; }
            else np[i] = (char)(np[i + 1] | np[n.c]);
        }
        return np;
    }

    void append_nodes(const BuiltBVH &bvh, uint32_t &node_base, const std::vector<char> *noprune = nullptr) {
        node_base = (uint32_t)out.nodes.size();
        for (const LinNode &n : bvh.nodes) {
            DNode d{};
            d.bmin[0] = n.b.min.x; d.bmin[1] = n.b.min.y; d.bmin[2] = n.b.min.z;
            d.bmax[0] = n.b.max.x; d.bmax[1] = n.b.max.y; d.bmax[2] = n.b.max.z;
            if (n.leaf) {
                // the reference tells leaves from interior nodes by n_primitives > 0 (bvh.rs:475); the walk relies on that too
                if ((n.c & 0xFFFFu) == 0u) throw Error("BVH build: leaf without primitives");
                d.link = n.a; d.meta = NODE_LEAF | (n.c & 0xFFFFu);
            }
            else { d.link = n.c; d.meta = n.a & 3u; }
            if (noprune && (*noprune)[out.nodes.size() - node_base]) d.meta |= NODE_NOPRUNE;
            out.nodes.push_back(d);
        }
    }

    // f64 -> f32 rounded toward -inf / +inf (a NaN stays a NaN; beyond the f32 range: the largest finite value or the infinity)
    static float f32_down(double x) {
        if (x != x) return (float)x;
        if (x > (double)FLT_MAX) return FLT_MAX;
        if (x < -(double)FLT_MAX) return -INFINITY;
        float f = (float)x;
        if ((double)f > x) f = std::nextafterf(f, -INFINITY);
        return f;
    }
    static float f32_up(double x) {
        if (x != x) return (float)x;
        if (x < -(double)FLT_MAX) return -FLT_MAX;
        if (x > (double)FLT_MAX) return INFINITY;
        float f = (float)x;
        if ((double)f < x) f = std::nextafterf(f, INFINITY);
        return f;
    }
    // Wide records (DNode4, dscene.h) of the FAST tree just appended at node_base: the root's two children, then the interior child
    // with the largest box opened in place until WIDE children stand (or only leaves are left); every interior child gets a record
    // of its own the same way.  Returns the wide walk's worst-case stack use: at a record with k children up to k - 1 are pending
    // while the walk is below the remaining one (extra_in_order[slot] = what a nested accel in that leaf slot needs on top).
    uint32_t wide_records(const BuiltBVH &bvh, uint32_t node_base, const std::vector<uint32_t> &extra_in_order) {
        out.nodes4.resize(out.nodes.size(), DNode4{});
        const std::vector<LinNode> &N = bvh.nodes;
        auto leaf_extra = [&](const LinNode &n) {
            uint32_t e = 0;
            for (uint32_t k = 0; k < (n.c & 0xFFFFu); ++k) e = std::max(e, extra_in_order[n.a + k]);
            return e;
        };
        if (N[0].leaf) return leaf_extra(N[0]);
        std::vector<char> made(N.size(), 0);
        std::vector<uint32_t> todo{0u};
        while (!todo.empty()) {
            const uint32_t i = todo.back(); todo.pop_back();
            made[i] = 1;
            std::vector<uint32_t> kids{i + 1u, N[i].c};
            while (kids.size() < (size_t)WIDE) {
                int pick = -1;
                double area = -1.0;
                for (size_t k = 0; k < kids.size(); ++k) {
                    if (N[kids[k]].leaf) continue;
                    const double a = b_area(N[kids[k]].b);
                    if (pick < 0 || a > area) { pick = (int)k; area = a; } // (a NaN area never replaces the pick)
                }
                if (pick < 0) break;
                const uint32_t c = kids[(size_t)pick];
                kids[(size_t)pick] = c + 1u;
                kids.insert(kids.begin() + pick + 1, N[c].c);
            }
            DNode4 &w = out.nodes4[node_base + i];
            for (int k = 0; k < WIDE; ++k) {
                w.link[k] = NO_HIT;
                for (int a = 0; a < 6; ++a) w.box[k][a] = 0.0f;
            }
            for (size_t k = 0; k < kids.size(); ++k) {
                const LinNode &c = N[kids[k]];
                const DNode &d = out.nodes[node_base + kids[k]];
                for (int a = 0; a < 3; ++a) { w.box[k][a] = f32_down(d.bmin[a]); w.box[k][3 + a] = f32_up(d.bmax[a]); }
                if (c.leaf) {
                    const uint32_t cnt = c.c & 0xFFFFu;
                    if (cnt > 7u || c.a > WIDE_START_MASK) throw Error("fast tree: leaf does not fit a wide record's link word");
                    w.link[k] = WIDE_LEAF | (cnt << WIDE_COUNT_SHIFT) | c.a;
                } else {
                    w.link[k] = kids[k];
                    todo.push_back(kids[k]);
                }
            }
        }
        std::vector<uint32_t> need(N.size(), 0u);
        for (size_t i = N.size(); i-- > 0;) { // children follow their parent in the linear order
            if (!made[i]) continue;
            const DNode4 &w = out.nodes4[node_base + i];
            uint32_t k = 0, below = 0;
            for (int c = 0; c < WIDE; ++c) {
                if (w.link[c] == NO_HIT) continue;
                ++k;
                if (w.link[c] & WIDE_LEAF) {
                    const uint32_t start = w.link[c] & WIDE_START_MASK, cnt = (w.link[c] >> WIDE_COUNT_SHIFT) & 7u;
                    for (uint32_t j = 0; j < cnt; ++j) below = std::max(below, extra_in_order[start + j]);
                }
                else below = std::max(below, need[w.link[c]]);
            }
            need[i] = k - 1u + below;
        }
        return need[0];
    }
    // parents of the reference tree just appended at node_base (fast mode's candidate check walks leaf -> root)
    void record_parents(const BuiltBVH &bvh, uint32_t node_base) {
        for (size_t i = 0; i < bvh.nodes.size(); ++i) out.nodes[node_base + i].parent = NO_HIT;
        for (size_t i = 0; i < bvh.nodes.size(); ++i) {
            const LinNode &n = bvh.nodes[i];
            if (n.leaf) continue;
            out.nodes[node_base + i + 1].parent = (uint32_t)i;
            out.nodes[node_base + n.c].parent = (uint32_t)i;
        }
    }
    // reference leaf of every primitive (NO_HIT for primitives the reference never tests: beyond a leaf's u16 count)
    static std::vector<uint32_t> ref_leaf_of_prims(const BuiltBVH &bvh, size_t nprims) {
        std::vector<uint32_t> leaf(nprims, NO_HIT);
        for (size_t i = 0; i < bvh.nodes.size(); ++i) {
            const LinNode &n = bvh.nodes[i];
            if (!n.leaf) continue;
            for (uint32_t k = 0; k < (n.c & 0xFFFFu); ++k) leaf[bvh.order[n.a + k]] = (uint32_t)i;
        }
        return leaf;
    }
    // boxes of the fast tree: the primitive boxes pushed out by 1e-9 of the accel's extent, so that a ray the reference
    // tree lets through to a primitive (its leaf boxes are unions, hence looser) is not lost at a tight fast-tree box
    // by a rounding-sized margin; what the looser boxes let through in excess is removed by the candidate check
    static std::vector<Bounds> inflated(const std::vector<Bounds> &pb) {
        double ext = 0.0;
        for (const Bounds &b : pb)
            for (int a = 0; a < 3; ++a) {
                double lo = std::fabs(comp(b.min, a)), hi = std::fabs(comp(b.max, a));
                if (std::isfinite(lo) && lo > ext) ext = lo;
                if (std::isfinite(hi) && hi > ext) ext = hi;
            }
#ifdef LG_NO_INFLATE
        const double e = 0.0 * ext;
#else
        const double e = ext * 1e-9 + 1e-300;
#endif
        std::vector<Bounds> r = pb;
        for (Bounds &b : r) { b.min = b.min - V3{e, e, e}; b.max = b.max + V3{e, e, e}; }
        return r;
    }

    void dump(const BuiltBVH &bvh, bool has_mat, bool swap, const Transform &t) {
        auto &f = out.dump_f; auto &i = out.dump_i;
        f.reserve(f.size() + bvh.nodes.size() * 6 + 32); i.reserve(i.size() + bvh.nodes.size() * 3 + bvh.order.size() + 4);
        i.push_back((int64_t)bvh.nodes.size()); i.push_back((int64_t)bvh.order.size());
        i.push_back(has_mat ? 1 : 0); i.push_back(swap ? 1 : 0);
        for (const LinNode &n : bvh.nodes) {
            f.push_back(n.b.min.x); f.push_back(n.b.min.y); f.push_back(n.b.min.z);
            f.push_back(n.b.max.x); f.push_back(n.b.max.y); f.push_back(n.b.max.z);
            i.push_back(n.leaf ? 1 : 0); i.push_back(n.a); i.push_back(n.c);
        }
        for (uint32_t o : bvh.order) i.push_back((int64_t)o);
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) f.push_back(t.m.m[c][r]);
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) f.push_back(t.minv.m[c][r]);
    }

    MeshTables &mesh_tables(uint32_t id) { // BVHAccel::from_mesh (bvh.rs:141-148), built once per mesh
        if (id >= scene.meshes.size()) throw Error("mesh handle out of range (the reference panics, bvh.rs:142)");
        MeshTables &mt = meshes[id];
        if (mt.built) return mt;
        const Obj &obj = *scene.meshes[id];
        size_t nf = obj.tri.size() / 3;
        auto tnow = [] { return std::chrono::steady_clock::now(); };
        auto tms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto ts0 = tnow();
        mt.has_n = !obj.normal.empty();
        mt.has_uv = !obj.texture.empty();
        uint32_t vbase = (uint32_t)(out.vpos.size() / 3), nbase = (uint32_t)(out.vnorm.size() / 3), tbase = (uint32_t)(out.vtex.size() / 2);
        out.vpos.insert(out.vpos.end(), obj.position.begin(), obj.position.end());
        out.vnorm.insert(out.vnorm.end(), obj.normal.begin(), obj.normal.end());
        out.vtex.insert(out.vtex.end(), obj.texture.begin(), obj.texture.end());
        mt.tri_base = (uint32_t)(out.tri_v.size() / 3);
        std::vector<Bounds> pb(nf);
        {
            const size_t at = out.tri_v.size();
            out.tri_v.resize(at + 3 * nf); out.tri_n.resize(at + 3 * nf); out.tri_t.resize(at + 3 * nf);
            uint32_t *tv = out.tri_v.data() + at, *tn = out.tri_n.data() + at, *tt = out.tri_t.data() + at;
            const bool has_n = mt.has_n, has_uv = mt.has_uv;
            for (size_t f = 0; f < nf; ++f) {
                V3 p[3];
                for (int k = 0; k < 3; ++k) {
                    const Obj::Tuple &tp = obj.tri[3 * f + k];
                    if (has_n && tp.n < 0) throw Error("mesh has normals but a face lacks a vn index (the reference panics, triangle.rs:60)");
                    if (has_uv && tp.t < 0) throw Error("mesh has vt but a face lacks a vt index (the reference panics, triangle.rs:96)");
                    const float *v = &obj.position[3 * (size_t)tp.v];
                    p[k] = V3{(double)v[0], (double)v[1], (double)v[2]};
                    tv[3 * f + k] = vbase + tp.v;
                    tn[3 * f + k] = has_n ? nbase + (uint32_t)tp.n : 0u;
                    tt[3 * f + k] = has_uv ? tbase + (uint32_t)tp.t : 0u;
                }
                pb[f] = b_add_point(b_new(p[0], p[1]), p[2]); // triangle.rs:157-159
            }
        }
        const auto ts1 = tnow();
        mt.bvh = Builder(pb, nf).run();
        const auto ts2 = tnow();
        mesh_build_ms += tms(ts1, ts2);
        append_nodes(mt.bvh, mt.node_base);
        record_parents(mt.bvh, mt.node_base);
        const std::vector<uint32_t> ref_leaf = ref_leaf_of_prims(mt.bvh, nf);
        out.tri_ref_leaf.resize((size_t)mt.tri_base + nf, NO_HIT);
        for (size_t f = 0; f < nf; ++f) out.tri_ref_leaf[mt.tri_base + f] = ref_leaf[f];
        mt.prim_base = (uint32_t)out.primref.size();
        {
            const size_t no = mt.bvh.order.size();
            out.primref.resize(mt.prim_base + no);
            out.leaf_soup.resize(mt.prim_base + no, DLeafRec{}); // slot j of the soup is aligned with primref[j]
            uint32_t *pr = out.primref.data() + mt.prim_base;
            DLeafRec *soup = out.leaf_soup.data() + mt.prim_base;
            const uint32_t tri_base = mt.tri_base;
            const uint32_t *order = mt.bvh.order.data();
            for (size_t j = 0; j < no; ++j) {
                const uint32_t o = order[j];
                pr[j] = (PK_TRIANGLE << 30) | (tri_base + o);
                // leaf-ordered copy of the three positions (f32, exactly the table entries), padded to 48 B
                for (int k = 0; k < 3; ++k)
                    std::memcpy(&soup[j].w[3 * k], &obj.position[3 * (size_t)obj.tri[3 * (size_t)o + k].v], 12);
            }
        }
        if (std::getenv("LASGUN_DEBUG_TIMES"))
            std::fprintf(stderr, "[lasgun] mesh %u (%zu triangles): vertex + index tables, boxes %.3f ms, BVH %.3f ms, nodes + slots + leaf soup %.3f ms\n", id, nf, tms(ts0, ts1), tms(ts1, ts2), tms(ts2, tnow()));
        mt.nnodes = (uint32_t)mt.bvh.nodes.size();
        mt.norder = (uint32_t)mt.bvh.order.size();
        mt.root_bounds = mt.bvh.nodes[0].b;
        mt.cmax = bounds_cmax(pb);
        std::vector<uint32_t> refs(mt.bvh.order.size(), 0);
        mt.max_stack = stack_need(mt.bvh.nodes, refs, [](uint32_t) { return 0u; });
        if (with_fast) { // fast tree over the same triangles
            const std::vector<Bounds> pbf = inflated(pb);
            BuiltBVH fb = FastBuilder(pbf, 1).run();
            append_nodes(fb, mt.fnode_base);
            mt.fprim_base = (uint32_t)out.primref.size();
            out.leaf_soup.resize(mt.fprim_base, DLeafRec{});
            for (uint32_t o : fb.order) {
                out.primref.push_back((PK_TRIANGLE << 30) | (mt.tri_base + o));
                DLeafRec rec{};
                for (int k = 0; k < 3; ++k)
                    std::memcpy(&rec.w[3 * k], &obj.position[3 * (size_t)obj.tri[3 * (size_t)o + k].v], 12);
                out.leaf_soup.push_back(rec);
            }
            mt.fmax_stack1 = wide_records(fb, mt.fnode_base, refs); // the wide walk: up to WIDE - 1 pending children per record
        }
        mt.built = true;
        return mt;
    }

    // minv is EXACTLY the identity (1.0 and +0.0 bit patterns): the traversal may then skip inverse_transform_ray
    // for rays whose components are finite and not -0 (walk.h, traverse_ref: the product is the same bits)
    static bool affine_is_identity(const Affine &m) {
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 3; ++r) {
                uint64_t bits;
                std::memcpy(&bits, &m.c[c][r], 8);
                if (bits != (c == r ? 0x3FF0000000000000ull : 0ull)) return false;
            }
        return true;
    }
    void set_chain(DAccel &a, int32_t parent, uint32_t self) {
        a.parent = parent;
        if (parent < 0) { a.nchain = 1; a.chain[0] = self; return; }
        const DAccel &p = out.accels[parent];
        if (p.nchain >= (uint32_t)MAX_CHAIN) throw Error("scene graph nested deeper than " + std::to_string(MAX_CHAIN) + " levels");
        a.nchain = p.nchain + 1;
        for (uint32_t i = 0; i < p.nchain; ++i) a.chain[i] = p.chain[i];
        a.chain[p.nchain] = self;
    }

    // returns accel id; sets bound = BVHAccel::bound() (bvh.rs:457-459) and need = stack entries
    uint32_t mesh_instance(uint32_t mesh, bool has_mat, const Material &mat, int32_t parent, Bounds &bound, uint32_t &need, uint32_t &fneed1) {
        uint32_t id = (uint32_t)out.accels.size();
        out.accels.emplace_back();
        MeshTables &mt = mesh_tables(mesh);
        Transform idt = transform_identity();
        DAccel a{};
        a.m = to_affine(idt.m); a.minv = to_affine(idt.minv);
        a.node_base = mt.node_base; a.prim_base = mt.prim_base;
        a.fnode_base = mt.fnode_base; a.fprim_base = mt.fprim_base;
        a.material = has_mat ? add_material(mat) : -1;
        a.flags = AF_MESH | (mt.has_n ? AF_HAS_N : 0u) | (mt.has_uv ? AF_HAS_UV : 0u) | (affine_is_identity(a.minv) ? AF_IDENTITY : 0u);
        set_chain(a, parent, id);
        prune_constants(mt.root_bounds, mt.cmax, 0.0, false, a.prune);
        out.accels[id] = a;
        dump(mt.bvh, has_mat, false, idt);
        bound = b_transform(idt.m, mt.root_bounds);
        need = mt.max_stack;
        fneed1 = mt.fmax_stack1;
        return id;
    }

    uint32_t aggregate(const Aggregate &agg, int32_t parent, Bounds &bound, uint32_t &need, uint32_t &fneed1) { // bvh.rs:150-162
        uint32_t id = (uint32_t)out.accels.size();
        out.accels.emplace_back();
        {
            DAccel a{};
            a.m = to_affine(agg.transform.m); a.minv = to_affine(agg.transform.minv);
            a.material = -1;
            a.flags = (agg.swap_backface ? AF_SWAP_BACKFACE : 0u) | (affine_is_identity(a.minv) ? AF_IDENTITY : 0u);
            set_chain(a, parent, id);
            out.accels[id] = a;
        }
        size_t n = agg.contents.size();
        if (n == 0) throw Error("empty aggregate: the reference recurses without bound in build_upper_sah (bvh.rs:355-424)");
        // The dump is pre-order (accel, then its child accels): reserve this accel's slot now.
        size_t dump_f_at = out.dump_f.size(), dump_i_at = out.dump_i.size();
        std::vector<double> child_f; std::vector<int64_t> child_i;
        std::swap(child_f, out.dump_f); std::swap(child_i, out.dump_i); // children dump into fresh vectors
        std::vector<Bounds> pb(n);
        std::vector<uint32_t> ref(n), extra(n, 0), fextra1(n, 0);
        std::vector<DLeafRec> rec(n, DLeafRec{});
        for (size_t i = 0; i < n; ++i) {
            const SceneNode &nd = agg.contents[i];
            switch (nd.kind) {
            case SceneNode::SPHERE: { // sphere.rs:19-25,73-77
                V3 c{nd.a[0], nd.a[1], nd.a[2]};
                double r = nd.b[0];
                out.spheres.push_back(DSphere{c.x, c.y, c.z, r});
                out.sphere_mat.push_back(add_material(nd.mat));
                pb[i] = b_new(c - V3{r, r, r}, c + V3{r, r, r});
                ref[i] = (PK_SPHERE << 30) | (uint32_t)(out.spheres.size() - 1);
                std::memcpy(rec[i].w, &out.spheres.back(), sizeof(DSphere));
                { const double r2 = r * r; std::memcpy(rec[i].w + 8, &r2, sizeof r2); } // rad*rad of Sphere::intersect_t (sphere.rs:52), the same IEEE product
                break;
            }
            case SceneNode::CUBE:   // cuboid.rs:24-30
            case SceneNode::CUBOID: { // cuboid.rs:18-22
                V3 p0{nd.a[0], nd.a[1], nd.a[2]};
                V3 p1 = nd.kind == SceneNode::CUBE ? p0 + V3{nd.b[0], nd.b[0], nd.b[0]} : V3{nd.b[0], nd.b[1], nd.b[2]};
                Bounds b = b_new(p0, p1);
                DCuboid dc{{b.min.x, b.min.y, b.min.z}, {b.max.x, b.max.y, b.max.z}};
                out.cuboids.push_back(dc);
                out.cuboid_mat.push_back(add_material(nd.mat));
                pb[i] = b;
                ref[i] = (PK_CUBOID << 30) | (uint32_t)(out.cuboids.size() - 1);
                std::memcpy(rec[i].w, &dc, sizeof(DCuboid));
                break;
            }
            case SceneNode::MESH: {
                uint32_t cn = 0, fcn1 = 0;
                uint32_t cid = mesh_instance(nd.obj, nd.has_mat, nd.mat, (int32_t)id, pb[i], cn, fcn1);
                fextra1[i] = 3 + fcn1;
                ref[i] = (PK_ACCEL << 30) | cid;
                extra[i] = 3 + cn;
                break;
            }
            case SceneNode::GROUP: {
                uint32_t cn = 0, fcn1 = 0;
                uint32_t cid = aggregate(*nd.group, (int32_t)id, pb[i], cn, fcn1);
                fextra1[i] = 3 + fcn1;
                ref[i] = (PK_ACCEL << 30) | cid;
                extra[i] = 3 + cn;
                break;
            }
            }
        }
        BuiltBVH bvh = Builder(pb, n).run();
        uint32_t node_base;
        const std::vector<char> over_accels = nodes_over_accels(bvh, ref);
        append_nodes(bvh, node_base, &over_accels);
        record_parents(bvh, node_base);
        {
            double rmin = INFINITY;
            bool has_sphere = false;
            for (size_t i = 0; i < n; ++i)
                if (agg.contents[i].kind == SceneNode::SPHERE) { has_sphere = true; const double r = agg.contents[i].b[0]; if (!(r >= rmin)) rmin = r; }
            prune_constants(bvh.nodes[0].b, bounds_cmax(pb), rmin, has_sphere, out.accels[id].prune);
        }
        const std::vector<uint32_t> ref_leaf = ref_leaf_of_prims(bvh, n);
        out.accel_ref_leaf.resize(out.accels.size(), NO_HIT);
        out.sphere_ref_leaf.resize(out.spheres.size(), NO_HIT);
        out.cuboid_ref_leaf.resize(out.cuboids.size(), NO_HIT);
        for (size_t i = 0; i < n; ++i) {
            const uint32_t kind = ref[i] >> 30, idx = ref[i] & PRIM_INDEX_MASK;
            if (kind == PK_ACCEL) out.accel_ref_leaf[idx] = ref_leaf[i];
            else if (kind == PK_SPHERE) out.sphere_ref_leaf[idx] = ref_leaf[i];
            else if (kind == PK_CUBOID) out.cuboid_ref_leaf[idx] = ref_leaf[i];
        }
        uint32_t prim_base = (uint32_t)out.primref.size();
        std::vector<uint32_t> extra_in_order(bvh.order.size());
        out.leaf_soup.resize(prim_base, DLeafRec{});
        for (size_t i = 0; i < bvh.order.size(); ++i) {
            out.primref.push_back(ref[bvh.order[i]]);
            out.leaf_soup.push_back(rec[bvh.order[i]]);
            extra_in_order[i] = extra[bvh.order[i]];
        }
        out.accels[id].node_base = node_base;
        out.accels[id].prim_base = prim_base;
        need = stack_need(bvh.nodes, extra_in_order, [](uint32_t e) { return e; });
        bound = b_transform(agg.transform.m, bvh.nodes[0].b);
        fneed1 = 0;
        if (with_fast) { // fast tree over the same primitives (child accels included as primitives)
            const std::vector<Bounds> pbf = inflated(pb);
            BuiltBVH fb = FastBuilder(pbf, 1).run();
            uint32_t fnode_base;
            append_nodes(fb, fnode_base);
            uint32_t fprim_base = (uint32_t)out.primref.size();
            out.leaf_soup.resize(fprim_base, DLeafRec{});
            std::vector<uint32_t> one(fb.order.size()); // per slot: what a nested accel there needs on the stack (3-word level frame + its own walk)
            for (size_t i = 0; i < fb.order.size(); ++i) {
                out.primref.push_back(ref[fb.order[i]]);
                out.leaf_soup.push_back(rec[fb.order[i]]);
                one[i] = fextra1[fb.order[i]];
            }
            out.accels[id].fnode_base = fnode_base;
            out.accels[id].fprim_base = fprim_base;
            fneed1 = wide_records(fb, fnode_base, one);
        }
        // stitch the dump: [prefix][this accel][children]
        std::swap(child_f, out.dump_f); std::swap(child_i, out.dump_i); // out.* = prefix again, child_* = children
        (void)dump_f_at; (void)dump_i_at;
        dump(bvh, false, agg.swap_backface, agg.transform);
        out.dump_f.insert(out.dump_f.end(), child_f.begin(), child_f.end());
        out.dump_i.insert(out.dump_i.end(), child_i.begin(), child_i.end());
        return id;
    }
};
} // namespace

// ---- culling records of the pruned walk (DChunk) --------------------------------------------------------
// A fat leaf of a mesh's reference tree is cut into RUNS of <= 32 triangles that are neighbours in space and face the same way
// (the reference's own order inside a leaf is a Morton order that ignores x, bvh.rs:575-579; a leaf is typically two to four
// separate patches of the surface).  Recursive splitting of the leaf's triangle set: at the largest gap between centroids along
// the widest axis when there is a clear one (patches fall apart there), at the median otherwise; a set of <= 32 triangles whose
// normals stay within 35 degrees of their mean is a run.
struct LeafTri {
    uint32_t slot; // its slot in leaf_soup
    V3 cen, n;     // centroid, unit normal (zero for a degenerate triangle)
};
constexpr size_t CHUNK_MIN_RUN = 2; // sets this small are never split further
static void cut_runs(std::vector<LeafTri> &t, size_t a, size_t b, std::vector<std::pair<size_t, size_t>> &runs) {
    const size_t count = b - a;
    if (count == 0) return;
    if (count <= ((size_t)1 << CHUNK_SHIFT)) {
        V3 sum{0, 0, 0};
        for (size_t i = a; i < b; ++i) { V3 n = t[i].n; if (dot(n, t[a].n) < 0.0) n = V3{-n.x, -n.y, -n.z}; sum = sum + n; }
        const double l = std::sqrt(dot(sum, sum));
        double cmin = 1.0;
        if (l > 0.0) for (size_t i = a; i < b; ++i) cmin = std::fmin(cmin, std::fabs(dot(t[i].n, sum)) / l);
        // (cos 35 degrees.  A set of <= CHUNK_MIN_RUN triangles is a run whatever its normals do: make_record measures every
        // run's cone itself and leaves lateral culling off when it is too wide, so this only bounds the number of records)
        if (count <= CHUNK_MIN_RUN || (l > 0.0 && cmin >= 0.82)) { runs.emplace_back(a, b); return; }
    }
    V3 lo = t[a].cen, hi = lo;
    for (size_t i = a; i < b; ++i) {
        const V3 c = t[i].cen;
        lo = V3{std::fmin(lo.x, c.x), std::fmin(lo.y, c.y), std::fmin(lo.z, c.z)};
        hi = V3{std::fmax(hi.x, c.x), std::fmax(hi.y, c.y), std::fmax(hi.z, c.z)};
    }
    const V3 e = hi - lo;
    const int axis = e.x >= e.y ? (e.x >= e.z ? 0 : 2) : (e.y >= e.z ? 1 : 2);
    std::sort(t.begin() + (long)a, t.begin() + (long)b, [&](const LeafTri &p, const LeafTri &q) {
        const double x = comp(p.cen, axis), y = comp(q.cen, axis);
        return x < y || (x == y && p.slot < q.slot);
    });
    size_t cut = a + count / 2;
    const double extent = comp(e, axis);
    if (extent > 0.0) {
        double gap = 0.0;
        size_t at = cut;
        for (size_t i = a + 1; i < b; ++i) {
            const double g = comp(t[i].cen, axis) - comp(t[i - 1].cen, axis);
            if (g > gap) { gap = g; at = i; }
        }
        if (gap > 4.0 * extent / (double)count) cut = at; // a clear gap: two patches
    }
    cut_runs(t, a, cut, runs);
    cut_runs(t, cut, b, runs);
}
// One culling record over the triangles in leaf_soup2 slots [a, b): bounds, normal cone, shape numbers.
static DChunk make_record(const FlatScene &out, size_t a, size_t b) {
    DChunk k{};
    k.start = (uint32_t)a;
    k.count = (uint32_t)(b - a);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    double g2 = 0.0, hmin = INFINITY;
    std::vector<V3> normals; // (one allocation, sized for the record: a run or a group of runs)
    normals.reserve(b - a);
    bool degenerate = false;
    V3 nsum{0, 0, 0};
    for (size_t s = a; s < b; ++s) {
        float p[9];
        std::memcpy(p, out.leaf_soup2[s].w, sizeof p);
        V3 v[3];
        for (int i = 0; i < 3; ++i) {
            v[i] = V3{(double)p[3 * i], (double)p[3 * i + 1], (double)p[3 * i + 2]};
            for (int ax = 0; ax < 3; ++ax) { mn[ax] = std::fmin(mn[ax], p[3 * i + ax]); mx[ax] = std::fmax(mx[ax], p[3 * i + ax]); }
        }
        const V3 e0 = v[1] - v[0], e1 = v[2] - v[1], e2 = v[0] - v[2];
        const V3 n = cross(e0, V3{-e2.x, -e2.y, -e2.z});
        const double twice_area = std::sqrt(dot(n, n));
        const double lmax = std::sqrt(std::fmax(dot(e0, e0), std::fmax(dot(e1, e1), dot(e2, e2))));
        const double h = twice_area / lmax; // the smallest altitude
        if (!(twice_area > 0.0) || !std::isfinite(twice_area) || !(h > 0.0)) { degenerate = true; continue; }
        V3 nu = n * (1.0 / twice_area);
        if (!normals.empty() && dot(nu, normals[0]) < 0.0) nu = V3{-nu.x, -nu.y, -nu.z}; // only |n . d| matters: one hemisphere, whatever the winding
        normals.push_back(nu);
        nsum = nsum + nu;
        g2 = std::fmax(g2, lmax * lmax / (h * h * h));
        hmin = std::fmin(hmin, h);
    }
    for (int ax = 0; ax < 3; ++ax) { k.bmin[ax] = mn[ax]; k.bmax[ax] = mx[ax]; }
    k.cos_t = -1.0f; k.sin_t = 1.0f; k.g2 = INFINITY; k.hmin = 0.0f; // (-1 = no lateral culling for this record)
    k.axis[0] = k.axis[1] = 0.0f; k.axis[2] = 1.0f;
    const double nl = std::sqrt(dot(nsum, nsum));
    if (!normals.empty() && !degenerate && nl > 1e-6 && std::isfinite(g2)) {
        // the axis as the kernel will see it (f32, not exactly unit): the cone's half-angle is measured against THAT vector,
        // and the kernel's |axis . d| / |d| differs from the cosine by the axis's length (1 +- 1e-7): 1e-3 rad of slack covers both
        const float ax[3] = {(float)(nsum.x / nl), (float)(nsum.y / nl), (float)(nsum.z / nl)};
        const double al = std::sqrt((double)ax[0] * ax[0] + (double)ax[1] * ax[1] + (double)ax[2] * ax[2]);
        double cmin = 1.0; // smallest cosine between a normal and the axis
        for (const V3 &n : normals) cmin = std::fmin(cmin, (n.x * ax[0] + n.y * ax[1] + n.z * ax[2]) / al);
        const double theta = std::acos(std::fmax(-1.0, std::fmin(1.0, cmin))) + 1e-3;
        if (theta < 1.5) {
            k.axis[0] = ax[0]; k.axis[1] = ax[1]; k.axis[2] = ax[2];
            k.cos_t = std::nextafterf((float)(std::cos(theta) * (1.0 - 1e-6)), -INFINITY);
            k.sin_t = std::nextafterf((float)(std::sin(theta) * (1.0 + 1e-6)), INFINITY);
            k.g2 = std::nextafterf((float)(g2 * (1.0 + 1e-6)), INFINITY);
            k.hmin = std::nextafterf((float)(hmin * (1.0 - 1e-6)), 0.0f);
        }
    }
    return k;
}
// The triangles in leaf_soup2 slots [a, b) (one run, <= 32 of them) as triangle strips, appended to out.strips: greedy -- of all the
// strips an unused triangle could start (three rotations each; a strip keeps taking the unused triangle that contains the last two
// vertices emitted) the longest is emitted, until none is left.  Vertices are identified by their f32 bits (what the reference's transforms see); winding plays no
// part (the sign test the strips serve is indifferent to it).  Returns the number of entries.
static uint32_t make_strips(const FlatScene &out, size_t a, size_t b, std::vector<DStrip> &strips) {
    // (fixed-size storage: a run has <= 32 triangles, hence <= 96 distinct vertices and <= 32 triangles at any of them -- this function
    // is where the accel build of a mesh spends its time, and with vectors of vectors most of that was the allocator)
    constexpr size_t MAXT = (size_t)1 << CHUNK_SHIFT, MAXV = 3 * MAXT;
    const size_t n = b - a;
    if (n > MAXT) throw Error("internal: a run of more than 32 triangles");
    struct T { uint32_t raw[3][3]; int v[3]; uint32_t slot; bool used; };
    T t[MAXT];
    uint32_t verts[MAXV][3]; // the run's distinct vertices (by their f32 bits)
    size_t nverts = 0;
    for (size_t i = 0; i < n; ++i) {
        T &x = t[i];
        std::memcpy(x.raw, out.leaf_soup2[a + i].w, 36);
        x.slot = out.leaf_soup2[a + i].w[9];
        x.used = false;
        for (int k = 0; k < 3; ++k) {
            size_t id = 0;
            while (id < nverts && !(verts[id][0] == x.raw[k][0] && verts[id][1] == x.raw[k][1] && verts[id][2] == x.raw[k][2])) ++id;
            if (id == nverts) { verts[id][0] = x.raw[k][0]; verts[id][1] = x.raw[k][1]; verts[id][2] = x.raw[k][2]; ++nverts; }
            x.v[k] = (int)id;
        }
    }
    // who touches each vertex (a strip step looks for an unused triangle that holds the last two vertices)
    uint8_t at[MAXV][3 * MAXT], at_n[MAXV];
    for (size_t v = 0; v < nverts; ++v) at_n[v] = 0;
    for (size_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { const size_t v = (size_t)t[i].v[k]; at[v][at_n[v]++] = (uint8_t)i; }
    auto emit = [&](const uint32_t *p, uint32_t code) {
        DStrip e;
        std::memcpy(&e.x, p, 12);
        e.code = code;
        strips.push_back(e);
    };
    const size_t before = strips.size();
    char taken[MAXT];
    for (size_t i = 0; i < n; ++i) taken[i] = 0;
    struct Step { int tri, third; };
    // the triangles a strip started at triangle i0 in rotation rot would take, in order (nothing is marked for good)
    auto follow = [&](size_t i0, int rot, Step *path) -> size_t {
        size_t len = 0;
        taken[i0] = 1;
        int l0 = t[i0].v[(rot + 1) % 3], l1 = t[i0].v[(rot + 2) % 3];
        for (;;) {
            int next = -1, third = -1;
            for (size_t e = 0; e < at_n[(size_t)l0]; ++e) {
                const int j = at[(size_t)l0][e];
                if (t[(size_t)j].used || taken[(size_t)j]) continue;
                const T &y = t[(size_t)j];
                int p = -1, q = -1;
                for (int k = 0; k < 3; ++k) { if (y.v[k] == l0 && p < 0) p = k; }
                for (int k = 0; k < 3; ++k) { if (y.v[k] == l1 && k != p && q < 0) q = k; }
                if (p >= 0 && q >= 0) { next = j; third = 3 - p - q; break; }
            }
            if (next < 0) break;
            path[len++] = Step{next, third};
            taken[(size_t)next] = 1;
            l0 = l1;
            l1 = t[(size_t)next].v[third];
        }
        taken[i0] = 0;
        for (size_t e = 0; e < len; ++e) taken[(size_t)path[e].tri] = 0;
        return len;
    };
    static const bool exhaustive = std::getenv("LASGUN_STRIPS_EXHAUSTIVE") != nullptr;
    size_t left = n;
    Step path[MAXT], best_path[MAXT];
    while (left != 0) {
        // the longest strip any unused triangle can start (ties: the first in run order, the lowest rotation)
        size_t bi = n, best_len = 0;
        int brot = 0;
        // candidates: every unused triangle (LASGUN_STRIPS_EXHAUSTIVE: 1.32 entries per triangle on the 100k-triangle torus, at twice
        // the build time), or just the one with the fewest unused neighbours across its edges -- a corner of what is left, from which
        // the strips run along the patch instead of cutting it up (1.40; the first unused triangle: 1.52)
        size_t only = n;
        if (!exhaustive) {
            int best_deg = 4;
            for (size_t i0 = 0; i0 < n; ++i0) {
                if (t[i0].used) continue;
                int deg = 0;
                for (int k = 0; k < 3; ++k) {
                    const int u = t[i0].v[k], w = t[i0].v[(k + 1) % 3];
                    bool nb = false;
                    for (size_t e = 0; e < at_n[(size_t)u]; ++e) {
                        const int j = at[(size_t)u][e];
                        if ((size_t)j == i0 || t[(size_t)j].used) continue;
                        const T &y = t[(size_t)j];
                        if (y.v[0] == w || y.v[1] == w || y.v[2] == w) { nb = true; break; }
                    }
                    deg += nb ? 1 : 0;
                }
                if (deg < best_deg) { best_deg = deg; only = i0; }
            }
        }
        for (size_t i0 = 0; i0 < n; ++i0) {
            if (t[i0].used || (only != n && i0 != only)) continue;
            for (int rot = 0; rot < 3; ++rot) {
                const size_t len = follow(i0, rot, path);
                if (bi == n || len > best_len) { bi = i0; brot = rot; best_len = len; std::copy(path, path + len, best_path); }
            }
        }
        T &x = t[bi];
        emit(x.raw[brot], 0u);
        emit(x.raw[(brot + 1) % 3], 0u);
        emit(x.raw[(brot + 2) % 3], STRIP_TRI | x.slot);
        x.used = true;
        --left;
        for (size_t e = 0; e < best_len; ++e) {
            const Step &st = best_path[e];
            emit(t[(size_t)st.tri].raw[st.third], STRIP_TRI | t[(size_t)st.tri].slot);
            t[(size_t)st.tri].used = true;
            --left;
        }
    }
    return (uint32_t)(strips.size() - before);
}
// leaf_soup2 (the triangles of every mesh leaf again, run after run; word 9 of a record = the slot it came from), the runs'
// records, and in DNode::pad of every mesh leaf: index of its first record | number of its records << 24.
static void build_chunks(FlatScene &out) {
    out.chunks.clear();
    out.leaf_soup2.clear();
    out.strips.clear();
    bool any_mesh = false;
    for (const DAccel &A : out.accels) any_mesh = any_mesh || (A.flags & AF_MESH) != 0u;
    if (!any_mesh) return; // (the pruned walk reads leaf_soup2 / chunks in mesh leaves only; capi.cpp points them at leaf_soup then)
    out.leaf_soup2 = out.leaf_soup;
    // every mesh leaf once (instances share their trees), in the order the records are laid out in
    struct LeafJob { uint32_t node; size_t first, count; std::vector<DChunk> chunks; std::vector<DStrip> strips; };
    std::vector<LeafJob> jobs;
    {
        std::vector<char> done(out.nodes.size(), 0);
        for (const DAccel &A : out.accels) {
            if (!(A.flags & AF_MESH) || done[A.node_base]) continue;
            std::vector<uint32_t> todo{0};
            while (!todo.empty()) {
                const uint32_t nidx = todo.back(); todo.pop_back();
                done[A.node_base + nidx] = 1;
                const DNode &nd = out.nodes[A.node_base + nidx];
                if (!(nd.meta & NODE_LEAF)) { todo.push_back(nidx + 1); todo.push_back(nd.link); continue; }
                jobs.push_back(LeafJob{A.node_base + nidx, (size_t)A.prim_base + nd.link, (size_t)(nd.meta & 0xFFFFu), {}, {}});
            }
        }
    }
    // A leaf's runs, records and strips depend on that leaf alone (its own slots of leaf_soup / leaf_soup2): the leaves are worked on by
    // several host threads, each into vectors of the leaf's own (record `pad` words relative to the leaf's first strip entry), and laid
    // out one after the other below -- the tables are what one thread makes, whatever the thread count.  (The strips doubled the
    // time of this function; lg_capture rebuilds the accel on every call, as the reference does.)
    auto one_leaf = [&out](LeafJob &J) {
        const size_t first = J.first, count = J.count;
        std::vector<LeafTri> tris(count);
        for (size_t i = 0; i < count; ++i) {
            float p[9];
            std::memcpy(p, out.leaf_soup[first + i].w, sizeof p);
            const V3 v0{p[0], p[1], p[2]}, v1{p[3], p[4], p[5]}, v2{p[6], p[7], p[8]};
            V3 n = cross(v1 - v0, v2 - v0);
            const double l = std::sqrt(dot(n, n));
            n = l > 0.0 && std::isfinite(l) ? n * (1.0 / l) : V3{0, 0, 0};
            tris[i] = LeafTri{(uint32_t)(first + i), (v0 + v1 + v2) * (1.0 / 3.0), n};
        }
        std::vector<std::pair<size_t, size_t>> runs;
        cut_runs(tris, 0, count, runs);
        // The records are an optimisation: a leaf whose cut does not fit the 8-bit record count of DNode::pad (a soup of
        // incoherent triangles, a geometric progression of centroids: the gap rule peels one triangle per cut) is cut into
        // plain runs of <= 32 consecutive triangles of the sorted set instead (<= 8 runs + 4 group records), and a mesh
        // beyond the 24-bit record index leaves its remaining leaves without records: pad = 0, and the leaf loop walks
        // such a leaf in the reference's order over the reference's soup (walk.h, mesh_leaf2).
        auto records_of = [](size_t nruns) { return nruns + nruns / CHUNK_GROUP; }; // one record per run, one per group of CHUNK_GROUP (= 2) runs
        if (records_of(runs.size()) > 255) {
            runs.clear();
            for (size_t r0 = 0; r0 < count; r0 += (size_t)1 << CHUNK_SHIFT) runs.emplace_back(r0, std::min(count, r0 + ((size_t)1 << CHUNK_SHIFT)));
        }
        for (size_t i = 0; i < count; ++i) { // (this leaf's slots of leaf_soup2: no other leaf reads or writes them)
            DLeafRec r = out.leaf_soup[tris[i].slot];
            r.w[9] = tris[i].slot;
            out.leaf_soup2[first + i] = r;
        }
        // the leaf's record stream: runs in groups of <= CHUNK_GROUP, every group of two or more behind a GROUP record over the
        // triangles of all its runs (start = CHUNK_IS_GROUP, count = how many run records follow it: a culled group is
        // stepped over whole)
        for (size_t g = 0; g < runs.size(); g += CHUNK_GROUP) {
            const size_t ge = std::min(runs.size(), g + CHUNK_GROUP);
            if (ge - g >= 2) {
                DChunk gk = make_record(out, first + runs[g].first, first + runs[ge - 1].second);
                gk.start = CHUNK_IS_GROUP;
                gk.count = (uint32_t)(ge - g);
                J.chunks.push_back(gk);
            }
            for (size_t r = g; r < ge; ++r) {
                DChunk k = make_record(out, first + runs[r].first, first + runs[r].second);
                k.pad = (uint32_t)J.strips.size(); // (relative to the leaf's first entry until the tables are laid out)
                const uint32_t entries = make_strips(out, first + runs[r].first, first + runs[r].second, J.strips); // <= 3 * 32
                k.count |= entries << 8;
                J.chunks.push_back(k);
            }
        }
    };
    {
        size_t tris_total = 0;
        for (const LeafJob &J : jobs) tris_total += J.count;
        unsigned nthreads = std::thread::hardware_concurrency();
        if (const char *e = std::getenv("LASGUN_HOST_THREADS")) nthreads = (unsigned)std::max(1, std::atoi(e));
        nthreads = std::min({nthreads ? nthreads : 1u, 16u, (unsigned)(tris_total / 4096 + 1)}); // (a small mesh is not worth a thread's start)
        if (nthreads <= 1) { for (LeafJob &J : jobs) one_leaf(J); }
        else {
            std::atomic<size_t> next{0};
            std::exception_ptr failed;
            std::mutex failed_mtx;
            std::vector<std::thread> pool;
            for (unsigned w = 0; w < nthreads; ++w)
                pool.emplace_back([&] {
                    try { for (size_t j; (j = next.fetch_add(1)) < jobs.size();) one_leaf(jobs[j]); }
                    catch (...) { std::lock_guard<std::mutex> lk(failed_mtx); if (!failed) failed = std::current_exception(); }
                });
            for (std::thread &th : pool) th.join();
            if (failed) std::rethrow_exception(failed);
        }
    }
    for (LeafJob &J : jobs) {
        DNode &nd = out.nodes[J.node];
        if (J.chunks.size() > 255) throw Error("internal: a leaf's culling records exceed their 8-bit count"); // (cannot happen: see records_of above)
        if (out.chunks.size() + J.chunks.size() + 2 >= (1u << 24)) { nd.pad = 0u; continue; }
        if (out.strips.size() + J.strips.size() >= 0xFFFFFF00u) throw Error("too many strip entries"); // (cannot happen: < 3 entries per slot, < 80M slots)
        nd.pad = (uint32_t)out.chunks.size() | (uint32_t)J.chunks.size() << 24;
        const uint32_t strip_base = (uint32_t)out.strips.size();
        for (DChunk &k : J.chunks) { if (k.start != CHUNK_IS_GROUP) k.pad += strip_base; out.chunks.push_back(k); }
        out.strips.insert(out.strips.end(), J.strips.begin(), J.strips.end());
        std::vector<DChunk>().swap(J.chunks); std::vector<DStrip>().swap(J.strips);
    }
    if (std::getenv("LASGUN_DEBUG_CHUNKS")) { // what the runs look like: sizes, and how much room their cones leave
        size_t n = out.chunks.size(), never = 0, tris = 0, hist[6] = {0, 0, 0, 0, 0, 0}, sizes[5] = {0, 0, 0, 0, 0}, nruns = 0;
        for (const DChunk &k : out.chunks) {
            if (k.start != CHUNK_IS_GROUP) { const size_t c = k.count & 0xFFu; tris += c; ++nruns; ++sizes[c <= 4 ? 0 : c <= 8 ? 1 : c <= 16 ? 2 : c <= 24 ? 3 : 4]; }
            if (k.cos_t < 0.0f) { ++never; continue; }
            const double room = 90.0 - std::acos((double)k.cos_t) * 57.29578; // how far from the axis a ray may point before it is edge-on to some triangle
            ++hist[room < 30 ? 0 : room < 45 ? 1 : room < 60 ? 2 : room < 70 ? 3 : room < 80 ? 4 : 5];
        }
        std::fprintf(stderr, "[lasgun] culling records: %zu (%zu runs) over %zu triangles (%.1f per run; runs of <=4: %zu, <=8: %zu, <=16: %zu, <=24: %zu, <=32: %zu); no lateral culling: %zu; room <30: %zu, <45: %zu, <60: %zu, <70: %zu, <80: %zu, >=80: %zu\n",
                     n, nruns, tris, nruns ? (double)tris / (double)nruns : 0.0, sizes[0], sizes[1], sizes[2], sizes[3], sizes[4], never, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
    }
    out.chunks.resize(out.chunks.size() + 2, DChunk{});
    out.strips.resize(out.strips.size() + 2, DStrip{0.f, 0.f, 0.f, 0u}); // two spare entries: the leaf loop keeps the next entry in flight
}

void flatten_scene(const Scene &scene, FlatScene &out, bool with_fast, bool with_records) {
    out = FlatScene();
    Flattener fl{scene, out, {}, with_fast};
    out.has_fast = with_fast;
    fl.meshes.resize(scene.meshes.size());
    out.default_material = fl.add_material(material_default());
    Bounds b;
    uint32_t need = 0, fneed1 = 0;
    static const bool times = std::getenv("LASGUN_DEBUG_TIMES") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    fl.aggregate(*scene.root, -1, b, need, fneed1);
    const auto t1 = std::chrono::steady_clock::now();
    out.max_stack = need;
    out.max_stack_fast1 = fneed1;
    if (out.primref.size() > 80000000u) throw Error("too many primitive slots for the 32-bit record offsets of the triangle stream");
    out.leaf_soup.resize(out.primref.size() + 2, DLeafRec{}); // two spare records: the mesh leaf loop keeps the next slot in flight
    if (with_records) build_chunks(out);
    {   // (a scene without a mesh has no leaf that could carry records: nothing to build on demand -- host.h)
        bool any_mesh = false;
        for (const DAccel &A : out.accels) any_mesh = any_mesh || (A.flags & AF_MESH) != 0u;
        out.has_records = with_records || !any_mesh;
    }
    if (times) {
        const auto t2 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[lasgun] flatten: scene graph + BVHs %.3f ms (mesh BVH builds %.3f of it), culling records + strips %.3f ms\n", ms(t0, t1), fl.mesh_build_ms, ms(t1, t2));
    }
    out.boxes_finite = true;
    for (const DNode &nd : out.nodes)
        for (int k = 0; k < 3; ++k) // finite AND ordered: slab_intersects_sg takes the near / far plane from the ray's sign, which is the reference's min / max only for bmin <= bmax
            out.boxes_finite = out.boxes_finite && std::isfinite(nd.bmin[k]) && std::isfinite(nd.bmax[k]) && nd.bmin[k] <= nd.bmax[k];
    out.sphere_ref_leaf.resize(out.spheres.size(), NO_HIT);
    out.cuboid_ref_leaf.resize(out.cuboids.size(), NO_HIT);
    out.tri_ref_leaf.resize(out.tri_v.size() / 3, NO_HIT);
    out.accel_ref_leaf.resize(out.accels.size(), NO_HIT);
    for (const Light &l : scene.lights) {
        DLight d;
        std::memcpy(d.pos, l.pos, sizeof d.pos);
        std::memcpy(d.intensity, l.intensity, sizeof d.intensity);
        std::memcpy(d.falloff, l.falloff, sizeof d.falloff);
        out.lights.push_back(d);
    }
}

} // namespace lg
