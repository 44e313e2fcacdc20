// lasgun_amd/csrc/kcommon.h -- what every kernel file of the ray-trace path starts from: launch constants and the
// per-lane stack in dynamic LDS.  (kernels.hip was one 3,100-line translation unit until round 3; it is now headers of
// device code -- walk.h, shade.h -- and one .hip file per kernel organisation, compiled side by side.)
#pragma once
#include <hip/hip_runtime.h>

#include "dscene.h"
#include "trig.h"

#ifndef LG_TRAV_WAVES_PER_SIMD
#define LG_TRAV_WAVES_PER_SIMD 4 // register budget of the traversal-only kernels
#endif
#define LG_BLOCK 256 // threads per workgroup; also the per-entry stride (in dwords) of the LDS stacks
#ifndef LG_WAVES_PER_SIMD
#define LG_WAVES_PER_SIMD 4 // register budget: 512 / 4 = 128 VGPRs per lane
#endif
// LDSS (reference traversal only): one 1024-lane workgroup per CU with the scene's node / primref / sphere / cuboid tables
// copied into LDS behind the stacks (walk.h, load_node); otherwise 256-lane workgroups and L1 / L2.
#ifndef LG_LDSS_BLOCK
#define LG_LDSS_BLOCK 1024 // (768 lanes -- three waves per SIMD, 168 registers, no spills -- measured: headline passes 3.08 + 3.14 -> 3.48 + 3.75 ms)
#endif

namespace lg {
// Work tiles of a persistent kernel, claimed XCD by XCD.  The tile sequence of a launch (8x8 pixel tiles in row order, or 64
// consecutive rays / hits of a queue) is cut into 8 contiguous bands, one per XCD, each with a head word on a cache line of its
// own: neighbouring tiles -- neighbouring pixels, the same part of the scene -- are then walked by CUs that share one 4 MiB L2
// instead of being dealt over all eight, and 256 CUs no longer queue on one atomic (MI355X_MICROARCH.md: 0.3 us against 2.8 us a
// claim).  A wave whose band is exhausted moves on to the next band for good, so the launch still drains evenly.  Which XCD a
// wave runs on is read from the hardware (HW_REG_XCC_ID); it only decides where a tile is rendered, never what is rendered.
// (TILE_HEADS, TILE_HEAD_STRIDE, TILE_HEAD0, TILE_GONE, TILE_COUNTER_WORDS, NO_TILE: dscene.h, shared with the host)
//
// How a launch ENDS (round 4).  A persistent grid holds W waves (4,096 for the LDS-resident kernels), and a wave used to learn that nothing
// was left by failing on every head: W x 8 failed fetch-adds on eight words, 75-90 microseconds -- the whole cost of a 64 x 64 film, half of a
// 512 x 512 one.  Three rules remove them, none of which changes what is rendered (which wave renders a tile never matters):
//   * a launch of n < W items is claimed by the first n waves only (numbered wave-in-workgroup major: n different CUs and SIMDs);
//   * in a SMALL launch (n <= TILE_SMALL_LAUNCH x W) the last floor(W / 8) tiles of every band -- W tiles at most -- are "final": a wave that
//     has rendered one leaves without asking again.  Each wave takes at most one final tile, there are no more final tiles than waves, and a
//     wave that has not had one keeps claiming: every tile is taken, and a claim only fails for a wave that moves from an exhausted band to
//     the next.  (Not for big launches: waves that leave after their own band's last tiles no longer help a slower band out -- the headline's
//     shadow pass went from 3.2 to 4.5 ms.  And not from ONE head word instead of eight: 16 k claims on one word are 0.3 ms by themselves.)
//   * the first wave to find a band exhausted sets its bit in the TILE_GONE word, and a wave that has just failed on one head reads that word
//     and steps over the bands whose bit is set.  (LOOKING at the next HEAD words instead -- plain loads of the contended lines -- was
//     measured too: the headline's traversal passes went from 3.10 to 3.40 ms.)
// Cornell glass at 512^2, level by level: 0.92 -> 0.42 ms; the 1-sphere README scene at 640^2: 0.25 -> 0.10 ms; headline unchanged.
constexpr uint32_t TILE_SMALL_LAUNCH = 2u; // (8: one rank's share of the headline frame at 8 GPUs -- 32,768 tiles -- went from 0.98 to 1.18 ms: its bands are not equally heavy)
__device__ __forceinline__ uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; } // XCC_ID[3:0], register 20 (gfx942 / gfx950)
__device__ __forceinline__ uint32_t grid_waves() { return gridDim.x * (blockDim.x >> 6); }
__device__ __forceinline__ bool wave_has_work(unsigned long long items) {
    return (unsigned long long)((threadIdx.x >> 6) * gridDim.x + blockIdx.x) < items;
}
// the single-head form (word [0]; kernels whose tiles are long: the claim is not what they wait for)
__device__ __forceinline__ uint32_t claim_tile_single(uint32_t *counter, uint32_t ntiles, bool &final) {
    uint32_t tile = 0u;
    if ((threadIdx.x & 63u) == 0u) tile = atomicAdd(counter, 1u);
    tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
    const uint32_t w = grid_waves();
    final = ntiles <= TILE_SMALL_LAUNCH * w && (unsigned long long)tile + w >= ntiles;
    return tile < ntiles ? tile : NO_TILE;
}
// the banded form: `band` is the wave's state (start it at xcc_id(), `left` at TILE_HEADS); called by every lane of the wave, the same tile
// comes back in all; `final` comes back true for a final tile of a small launch
__device__ __forceinline__ uint32_t claim_tile(uint32_t *counter, uint32_t ntiles, uint32_t &band, uint32_t &left, bool &final) {
    uint32_t tile = NO_TILE, fin = 0u;
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t w = grid_waves(), tail = ntiles <= TILE_SMALL_LAUNCH * w ? w / TILE_HEADS : 0u;
        uint32_t gone = 0u;
        while (left != 0u) {
            const uint32_t lo = (uint32_t)(((unsigned long long)band * ntiles) / TILE_HEADS), hi = (uint32_t)(((unsigned long long)(band + 1u) * ntiles) / TILE_HEADS);
            if (!((gone >> band) & 1u)) {
                const uint32_t t = atomicAdd(counter + TILE_HEAD0 + band * TILE_HEAD_STRIDE, 1u);
                if (t < hi - lo) { tile = lo + t; fin = (t + tail >= hi - lo) ? 1u : 0u; break; }
                if (t == hi - lo) atomicOr(counter + TILE_GONE, 1u << band);
                gone = __hip_atomic_load(counter + TILE_GONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            band = (band + 1u) & (TILE_HEADS - 1u); // this band is done (for every wave: its head only grows)
            --left;
        }
    }
    band = (uint32_t)__builtin_amdgcn_readfirstlane((int)band);
    left = (uint32_t)__builtin_amdgcn_readfirstlane((int)left);
    final = __builtin_amdgcn_readfirstlane((int)fin) != 0;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
}
// claim_tile for a wave only SOME of whose lanes are here (the refilling walk, walk.h: the lanes whose ray is done): the leader is the first
// active lane -- which is also the lane readfirstlane reads -- and there are no "final" tiles (the caller runs big launches only).
__device__ __forceinline__ uint32_t claim_tile_partial(uint32_t *counter, uint32_t ntiles, uint32_t &band, uint32_t &left) {
    uint32_t tile = NO_TILE;
    const uint32_t leader = (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true));
    if ((threadIdx.x & 63u) == leader) {
        uint32_t gone = 0u;
        while (left != 0u) {
            const uint32_t lo = (uint32_t)(((unsigned long long)band * ntiles) / TILE_HEADS), hi = (uint32_t)(((unsigned long long)(band + 1u) * ntiles) / TILE_HEADS);
            if (!((gone >> band) & 1u)) {
                const uint32_t t = atomicAdd(counter + TILE_HEAD0 + band * TILE_HEAD_STRIDE, 1u);
                if (t < hi - lo) { tile = lo + t; break; }
                if (t == hi - lo) atomicOr(counter + TILE_GONE, 1u << band);
                gone = __hip_atomic_load(counter + TILE_GONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            band = (band + 1u) & (TILE_HEADS - 1u);
            --left;
        }
    }
    band = (uint32_t)__builtin_amdgcn_readfirstlane((int)band);
    left = (uint32_t)__builtin_amdgcn_readfirstlane((int)left);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
}
// A workgroup copies `n16` 16-byte units into its LDS: four loads in flight per lane before the first store.  (Written as the plain loop
// `dst[i] = src[i]` hipcc waits for every load before its store: ten dependent L2 round trips for the headline scene's 155 KB image,
// ~15 microseconds at the head of every traversal launch.)
__device__ __forceinline__ void copy_to_lds(uint4 *dst, const uint4 *src, uint32_t n16, uint32_t tid, uint32_t stride) {
    uint32_t i = tid;
    for (; i + 3u * stride < n16; i += 4u * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2u * stride], d = src[i + 3u * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2u * stride] = c; dst[i + 3u * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
} // namespace lg
