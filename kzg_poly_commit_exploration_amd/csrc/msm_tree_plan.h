// msm_tree_plan.h -- the work description of the log-depth tree sums (msm_reduce.hip), shared with the one-launch
// path of small jobs (msm_finalize.hip: k_small_msm).
#pragma once
#include "engine.h"

namespace kzg {

constexpr int kTreeBlock = 256;                 // threads per workgroup of the tree kernels
constexpr uint32_t kTreeLogical = kTreeBlock / 4;  // logical lanes (quads) per workgroup

// out[g] = sum_{q < len} in[g * gstride + q * estride]   for g < groups  (strides in XYZZ records).
// A workgroup of 256 lanes serves 256 / lanes_per_group groups; each lane first adds its share of the
// group serially (only when len > 256), then the lanes of a group fold in a tree through LDS.
// Up to four independent jobs share one launch (their dependent-addition chains run side by side).
struct TreeJob {
    const uint4* in;
    uint4* out;
    uint32_t groups, len, lanes_per_group, first_block, inner;
    uint64_t gstride, estride, ostride;
};

// fills j[0 .. count) and returns the number of workgroups they take; first_block = number of the first one
inline uint32_t plan_tree_jobs(TreeJob* j, const TreeSumDesc* descs, uint32_t count, uint32_t first_block, uint32_t waves_per_simd) {
    // One (logical) lane per element gives the shortest chain (log2(len) dependent additions) but only ~1/log2(len)
    // of the lane-steps do work.  A CU holds one workgroup (one wave per SIMD) = 64 logical lanes: beyond 256
    // workgroups a second round would start, which costs a whole tree's latency -- lanes then pre-add several
    // elements serially instead (one more dependent addition per doubling).
    uint64_t total = 0;
    for (uint32_t i = 0; i < count; i++) total += (uint64_t)descs[i].groups * descs[i].len;
    const uint64_t resident_lanes = (uint64_t)waves_per_simd * 256 * kTreeLogical;
    uint32_t per_lane = 1;
    while ((total + per_lane - 1) / per_lane > resident_lanes && per_lane < 64) per_lane <<= 1;
    uint32_t blocks = first_block;
    for (uint32_t i = 0; i < count; i++) {
        uint32_t lpg = 1;
        while (lpg < descs[i].len && lpg < kTreeLogical) lpg <<= 1;
        lpg = lpg / per_lane ? lpg / per_lane : 1;
        const uint32_t gpb = kTreeLogical / lpg;
        TreeJob& J = j[i];
        J.in = reinterpret_cast<const uint4*>(descs[i].in);
        J.out = reinterpret_cast<uint4*>(descs[i].out);
        J.groups = descs[i].groups;
        J.len = descs[i].len;
        J.lanes_per_group = lpg;
        J.first_block = blocks;
        J.gstride = descs[i].gstride;
        J.estride = descs[i].estride;
        J.inner = descs[i].inner ? descs[i].inner : descs[i].groups;
        J.ostride = descs[i].ostride;
        blocks += (descs[i].groups + gpb - 1) / gpb;
    }
    return blocks - first_block;
}

}  // namespace kzg
