// lasgun_amd/csrc/packet.h -- the packet walk: one tree traversal per wavefront (opt-in organisation, k_packet.hip).
#pragma once
#include "walk.h"

namespace lg {
// ------------------------------------------------------------------------------------------
// Packet traversal (streaming pipeline, reference tree): ONE walk per wavefront.
//
// The 64 rays of an 8x8 tile (primary) or of its hit points towards one light (shadow) visit
// almost the same nodes.  Instead of 64 private walks -- private node fetches, private stacks,
// private near/far selects -- the wave walks the UNION of its lanes' node sets once: the node and
// primitive records are fetched with wave-uniform addresses (one request, broadcast), the stack
// is one small per-wave array of (node, lane mask) in LDS, the near child is chosen by a vote of
// the lanes that hit the node (ballot + popcount on dir_is_neg[axis], bvh.rs:496), and a lane
// simply drops out of the mask at a node whose box it misses.
//
// Exactness.  A lane takes part in a primitive test iff every box on the path from the root to
// that leaf passed ITS OWN slab test -- exactly the reference's candidate set for that ray, since
// the reference never culls by t (cuboid.rs:120) -- and every test is the same arithmetic on the
// same operands.  The closest hit is the minimum of the accepted t over that set, which does not
// depend on the visiting order except (a) between candidates with exactly equal t, where the
// reference keeps the first one it visits, and (b) after a NaN t, which the reference's
// comparisons accept and which then accepts everything after it.  Both are detected per lane
// (`tie`) and that lane is re-traced with its private reference-order walk.  An occluded any-hit
// ray needs no re-trace: "some accepted t < 1 exists" is order-independent (NaN aside, as in
// the private walk).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ bool lane_in(unsigned long long m, uint32_t lane) { return ((m >> lane) & 1ull) != 0ull; }
constexpr uint32_t PKT_ENTRY = 4u; // dwords per wave-stack entry: {a, b, mask lo, mask hi}

__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    return (unsigned long long)uni((uint32_t)v) | ((unsigned long long)uni((uint32_t)(v >> 32)) << 32);
}
// Records of the packet walk: the same LDS image as the private walks (every lane reads the SAME address here,
// so its padding is irrelevant), or the tables in HBM / L2 with wave-uniform addresses: one request per record and wave.
template <bool LDSS>
__device__ __forceinline__ NodeRec pkt_node(const DParams &P, const uint4 *scn, uint32_t idx) { return load_node<LDSS>(P, scn, idx); }
struct SlotRec { // one leaf slot: its primref and its 48-byte geometry record
    uint32_t ref;
    LeafRec g;
};
template <bool LDSS>
__device__ __forceinline__ SlotRec pkt_slot(const DParams &P, const uint4 *scn, uint32_t slot, bool mesh, uint32_t soup_delta) {
    SlotRec r;
    if (LDSS) {
        r.ref = reinterpret_cast<const uint32_t *>(scn + P.lds_prim_off)[slot];
        if (mesh) { // triangle records are not part of the LDS image
            r.g = load_rec(P, slot + soup_delta);
        } else {
            const uint4 *q = scn + (P.lds_soup_off + slot * 3u);
            r.g = LeafRec{q[0], q[1], q[2]};
        }
    } else {
        r.ref = P.primref[slot];
        r.g = load_rec(P, slot);
    }
    return r;
}
// The wave-uniform state lives in plain locals and every value that comes back from memory goes through
// readfirstlane, so that the compiler keeps it in SGPRs and branches on it with scalar branches.  (No
// by-reference lambdas here: state that round-trips through a private-memory capture is treated as divergent.)
#define PKT_SET_LEVEL(ACC, LOCAL)                                                                                      \
    do {                                                                                                               \
        const DAccel *A_ = P.accels + (ACC);                                                                           \
        accel = (ACC);                                                                                                 \
        ray = (LOCAL);                                                                                                 \
        dd = dot(ray.d, ray.d);                                                                                        \
        node_base = uni(LDSS ? A_->lnode_base : A_->node_base);                                                        \
        prim_base = uni(LDSS ? A_->lprim_base : A_->prim_base);                                                        \
        soup_delta = uni(LDSS ? A_->prim_base - A_->lprim_base : 0u);                                                  \
        mesh = (uni(A_->flags) & AF_MESH) != 0u;                                                                       \
        negbits = (ray.dinv.x < 0.0 ? 1u : 0u) | (ray.dinv.y < 0.0 ? 2u : 0u) | (ray.dinv.z < 0.0 ? 4u : 0u);          \
    } while (0)
/* every lane stores the same four words to the same address: one LDS write, no exec juggling */
#define PKT_PUSH(SP, A, B, M)                                                                                          \
    do {                                                                                                               \
        *reinterpret_cast<uint4 *>(ws + (SP) * PKT_ENTRY) = uint4{(A), (B), (uint32_t)(M), (uint32_t)((M) >> 32)};     \
    } while (0)
#define PKT_MASK(SP) ((unsigned long long)uni(ws[(SP) * PKT_ENTRY + 2u]) | ((unsigned long long)uni(ws[(SP) * PKT_ENTRY + 3u]) << 32))
// one candidate of this lane: tie / NaN bookkeeping, acceptance, any-hit exit
#define PKT_CANDIDATE(VALID, T, REF)                                                                                   \
    do {                                                                                                               \
        if (VALID) {                                                                                                   \
            const double t_ = (T);                                                                                     \
            if ((t_ == best.t && best.ref != NO_HIT) || t_ != t_) tie = true;                                          \
            if (!(t_ >= best.t)) {                                                                                     \
                best.t = t_; best.ref = (REF); best.accel = accel;                                                     \
                if (anyhit && t_ < 1.0) alive = false; /* point.rs:49 */                                               \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)

template <bool LDSS>
__device__ __forceinline__ void traverse_packet(const DParams &P, const uint4 *scn, const Ray &wray, bool alive, const bool anyhit,
                                                uint32_t *ws, const uint32_t lane, Best &best, bool &tie) {
    best.t = INFINITY; best.ref = NO_HIT; best.accel = 0;
    tie = false;
    unsigned long long alive_m = __ballot(alive);
    if (alive_m == 0ull) return;
    // ---- level state: wave-uniform except the rays
    uint32_t accel = 0u, node_base = 0u, prim_base = 0u, soup_delta = 0u;
    bool mesh = false;
    Ray ray;
    double dd = 0.0;
    uint32_t negbits = 0u;                 // per lane: bit a set <=> dinv[a] < 0 (dir_is_neg, bvh.rs:463)
    const Ray root = ray_to_local(P.accels->minv, wray); // the world ray is not kept (see level_ray)
    const V3 root_o = root.o, root_d = root.d;
    PKT_SET_LEVEL(0u, root);
    uint32_t sp = 0u, base = 0u;           // wave stack, in entries
    uint32_t cur = 0u;                     // node to visit ...
    unsigned long long m = alive_m;        // ... by these lanes
    bool have = true;                      // (cur, m) is pending
    uint32_t li = 0u, le = 0u;             // leaf cursor (slots of this level's numbering) ...
    unsigned long long lm = 0ull;          // ... and the lanes inside the leaf
    bool leaf_open = false;
    for (;;) {
        // ---- phase A: interior nodes, one record fetch and one slab test per step for the whole wave
        // (readfirstlane pins: no-ops in hardware terms, they tell the compiler this state is wave-uniform)
        have = uni((uint32_t)have) != 0u; leaf_open = uni((uint32_t)leaf_open) != 0u;
        sp = uni(sp); base = uni(base); accel = uni(accel); node_base = uni(node_base); prim_base = uni(prim_base);
        alive_m = uni64(alive_m);
        while (have) {
            cur = uni(cur); m = uni64(m); sp = uni(sp);
            have = false;
            m &= alive_m;
            if (m == 0ull) break;
            const NodeRec nd = pkt_node<LDSS>(P, scn, node_base + cur);
            const uint32_t link = uni(nd.link), meta = uni(nd.meta);
            const bool inside_box = slab_intersects(nd.bmin, nd.bmax, ray); // every lane computes it: no exec juggling
            const unsigned long long hm = __ballot(inside_box) & m;
            if (hm == 0ull) break;
            if (meta & NODE_LEAF) {
                const uint32_t count = meta & 0xFFFFu;
                if (count != 0u) { li = prim_base + link; le = li + count; lm = hm; leaf_open = true; } // nprims as u16 == 0: nothing
                break;
            }
            const unsigned long long ng = __ballot(((negbits >> (meta & 3u)) & 1u) != 0u); // dir_is_neg[axis] of every lane
            const bool neg_first = 2 * __popcll(hm & ng) > __popcll(hm); // the vote: most lanes' near child first (bvh.rs:496)
            const uint32_t near_node = neg_first ? link : cur + 1u, far_node = neg_first ? cur + 1u : link;
            PKT_PUSH(sp, far_node, 0u, hm);
            ++sp;
            cur = near_node; m = hm; have = true;
        }
        // ---- phase B: the leaf's slots li .. le for the lanes lm, in order[] sequence, one slot ahead
        if (leaf_open) {
            leaf_open = false;
            li = uni(li); le = uni(le); lm = uni64(lm);
            lm &= alive_m;
            if (lm != 0ull) {
                TriSetup tri{2, 0.0, 0.0, 0.0};
                unsigned long long kz0 = 0ull, kz1 = 0ull, kz2 = 0ull;
                if (mesh) { // shear constants and dominant axis per fat leaf (as the private walk does)
                    tri = tri_setup(ray);
                    kz0 = __ballot(tri.kz == 0); kz1 = __ballot(tri.kz == 1); kz2 = __ballot(tri.kz == 2);
                }
                const uint32_t last = le - 1u;
                SlotRec nxt = pkt_slot<LDSS>(P, scn, li, mesh, soup_delta);
                while (li < le) {
                    li = uni(li); lm = uni64(lm);
                    const SlotRec s = nxt;
                    const uint32_t slot = li;
                    ++li;
                    nxt = pkt_slot<LDSS>(P, scn, li < last ? li : last, mesh, soup_delta); // prefetch (clamped: always a valid slot)
                    const uint32_t ref = uni(s.ref);
                    const uint32_t kind = ref >> 30, idx = ref & PRIM_INDEX_MASK;
                    const bool in = lane_in(lm, lane);
                    (void)slot;
                    if (kind == PK_TRIANGLE) {
                        const V3 p0{rec_f32(s.g.a.x), rec_f32(s.g.a.y), rec_f32(s.g.a.z)}, p1{rec_f32(s.g.a.w), rec_f32(s.g.b.x), rec_f32(s.g.b.y)},
                            p2{rec_f32(s.g.b.z), rec_f32(s.g.b.w), rec_f32(s.g.c.x)};
                        TriHit h;
                        if (mesh) { // the permutation is a per-ray property: one pass per dominant axis present among the leaf's lanes
                            if ((lm & kz0) != 0ull) { if (lane_in(lm & kz0, lane)) { const bool ok = triangle_t_pre<0>(p0, p1, p2, ray.o, tri.sx, tri.sy, tri.sz, h); PKT_CANDIDATE(ok, h.t, ref); } }
                            if ((lm & kz1) != 0ull) { if (lane_in(lm & kz1, lane)) { const bool ok = triangle_t_pre<1>(p0, p1, p2, ray.o, tri.sx, tri.sy, tri.sz, h); PKT_CANDIDATE(ok, h.t, ref); } }
                            if ((lm & kz2) != 0ull) { if (lane_in(lm & kz2, lane)) { const bool ok = triangle_t_pre<2>(p0, p1, p2, ray.o, tri.sx, tri.sy, tri.sz, h); PKT_CANDIDATE(ok, h.t, ref); } }
                        } else if (in) { // a triangle outside a mesh accel cannot be built by the scene API; kept for completeness
                            const bool ok = triangle_t(p0, p1, p2, ray, h);
                            PKT_CANDIDATE(ok, h.t, ref);
                        }
                    } else if (kind == PK_SPHERE) {
                        const V3 cen{rec_f64(s.g.a.x, s.g.a.y), rec_f64(s.g.a.z, s.g.a.w), rec_f64(s.g.b.x, s.g.b.y)};
                        const double rad = rec_f64(s.g.b.z, s.g.b.w);
                        // the discriminant for every lane first; the roots (sqrt, two divides) only if some lane of the leaf needs them
                        const V3 l = ray.o - cen;
                        const double b = 2.0 * dot(ray.d, l);
                        const double c = dot(l, l) - rad * rad;
                        const double disc = b * b - 4.0 * dd * c;
                        const bool need = in && (dd == 0.0 || !(disc < 0.0));
                        if (__ballot(need) != 0ull) {
                            if (need) {
                                bool inside;
                                const double t = sphere_t_a(ray, dd, cen, rad, inside);
                                PKT_CANDIDATE(!(t < 0.0), t, ref);
                            }
                        }
                    } else if (kind == PK_CUBOID) {
                        if (in) {
                            double mn[3] = {rec_f64(s.g.a.x, s.g.a.y), rec_f64(s.g.a.z, s.g.a.w), rec_f64(s.g.b.x, s.g.b.y)};
                            double mx[3] = {rec_f64(s.g.b.z, s.g.b.w), rec_f64(s.g.c.x, s.g.c.y), rec_f64(s.g.c.z, s.g.c.w)};
                            V3 d0, d1;
                            double t = 0.0;
                            const bool ok = cuboid_hit<false>(mn, mx, ray, t, d0, d1);
                            PKT_CANDIDATE(ok, t, ref);
                        }
                    } else { // PK_ACCEL -- nested BVHAccel: every lane of the leaf enters it (bvh.rs:483-488, 462)
                        PKT_PUSH(sp, li, le, lm);
                        PKT_PUSH(sp + 1u, base, 0u, 0ull);
                        sp += 2u; base = sp;
                        PKT_SET_LEVEL(idx, ray_to_local(P.accels[idx].minv, ray));
                        cur = 0u; m = lm; have = true;
                        break;
                    }
                    if (anyhit) {
                        alive_m = __ballot(alive);
                        lm &= alive_m;
                        if (lm == 0ull) break;
                    }
                }
            }
        }
        // ---- phase C: next pending (node, mask) of this level, or back to the parent's leaf
        if (!have) {
            if (sp != base) {
                --sp;
                cur = uni(ws[sp * PKT_ENTRY]); m = PKT_MASK(sp);
                have = true;
            } else {
                if (accel == 0u) break;
                sp -= 2u;                  // level frame: {li, le, leaf mask} {previous base}
                li = uni(ws[sp * PKT_ENTRY]); le = uni(ws[sp * PKT_ENTRY + 1u]); lm = PKT_MASK(sp);
                base = uni(ws[(sp + 1u) * PKT_ENTRY]);
                const uint32_t parent = uni((uint32_t)P.accels[accel].parent);
                PKT_SET_LEVEL(parent, level_ray(P, root_o, root_d, parent)); // recomputed, bit-identical to the first computation
                leaf_open = li < le;
            }
        }
    }
}
#undef PKT_SET_LEVEL
#undef PKT_PUSH
#undef PKT_MASK
#undef PKT_CANDIDATE


} // namespace lg
