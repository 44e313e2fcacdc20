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
#define LG_LDSS_BLOCK 1024

namespace lg {
// Work tiles of a persistent kernel, claimed XCD by XCD.  The tile sequence of a launch (8x8 pixel tiles in row order, or 64
// consecutive rays / hits of a queue) is cut into 8 contiguous bands, one per XCD, each with a head word on a cache line of its
// own: neighbouring tiles -- neighbouring pixels, the same part of the scene -- are then walked by CUs that share one 4 MiB L2
// instead of being dealt over all eight, and 256 CUs no longer queue on one atomic (MI355X_MICROARCH.md: 0.3 us against 2.8 us a
// claim).  A wave whose band is exhausted moves on to the next band for good, so the launch still drains evenly.  Which XCD a
// wave runs on is read from the hardware (HW_REG_XCC_ID); it only decides where a tile is rendered, never what is rendered.
// (TILE_HEADS, TILE_HEAD_STRIDE, TILE_COUNTER_WORDS, NO_TILE: dscene.h, shared with the host)
__device__ __forceinline__ uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; } // XCC_ID[3:0], register 20 (gfx942 / gfx950)
// `band` is the wave's state (start it at xcc_id(), `left` at TILE_HEADS); called by every lane of the wave, the same tile comes back in all
// (Measured and dropped, round 4: a wave that has found one band exhausted LOOKING at the next heads -- a plain load -- before it queues on
// them.  The headline's traversal passes went from 3.10 to 3.40 ms with it.)
__device__ __forceinline__ uint32_t claim_tile(uint32_t *counter, uint32_t ntiles, uint32_t &band, uint32_t &left) {
    uint32_t tile = NO_TILE;
    if ((threadIdx.x & 63u) == 0u) {
        while (left != 0u) {
            const uint32_t lo = (uint32_t)(((unsigned long long)band * ntiles) / TILE_HEADS), hi = (uint32_t)(((unsigned long long)(band + 1u) * ntiles) / TILE_HEADS);
            const uint32_t t = atomicAdd(counter + 16u + band * TILE_HEAD_STRIDE, 1u);
            if (t < hi - lo) { tile = lo + t; break; }
            band = (band + 1u) & (TILE_HEADS - 1u); // this band is done (for every wave: its head only grows)
            --left;
        }
    }
    band = (uint32_t)__builtin_amdgcn_readfirstlane((int)band);
    left = (uint32_t)__builtin_amdgcn_readfirstlane((int)left);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)tile);
}
// The persistent grids hold as many waves as the chip runs at once (4,096 for the LDS-resident kernels); a launch with fewer work
// items than that has nothing for the surplus, and each surplus wave would still queue once on every head word before it leaves --
// for a 64 x 64 film that queueing was the whole frame (75-90 microseconds a launch, measured round 4).  Waves are numbered
// wave-in-workgroup major, so the first n of them sit on n different CUs (and SIMDs): those claim, the others leave at once.  Which
// wave renders a tile never matters; that the claiming waves are all resident does -- the grid is sized so that every wave is.
__device__ __forceinline__ bool wave_has_work(unsigned long long items) {
    return (unsigned long long)((threadIdx.x >> 6) * gridDim.x + blockIdx.x) < items;
}
} // namespace lg
