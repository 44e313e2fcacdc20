// lasgun_amd/csrc/kcommon.h -- what every kernel file of the ray-trace path starts from: launch constants and the
// per-lane stack in dynamic LDS.  (kernels.hip was one 3,100-line translation unit until round 3; it is now headers of
// device code -- walk.h, packet.h, shade.h -- and one .hip file per kernel organisation, compiled side by side.)
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
