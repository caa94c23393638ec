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
inline uint32_t plan_tree_jobs(TreeJob* j, const TreeSumDesc* descs, uint32_t count, uint32_t first_block, uint32_t resident_quads) {
    // One (logical) lane per element gives the shortest chain (log2(len) dependent additions) but only ~1/log2(len)
    // of the lane-steps do work, and a group cannot be wider than one workgroup (64 quads).  A CU holds one workgroup
    // (one wave per SIMD): beyond 256 workgroups a second round would start, which costs a whole tree's latency, so
    // the groups are narrowed until the launch fits one round -- lanes then pre-add several elements serially (one more
    // dependent addition per halving).  (Round 2 fix: the first version divided the already capped width once more and
    // gave the 256-element sums of a 2^20-term job 8 quads of 32 serial additions each, on a quarter of the chip:
    // 285 us for the two stages.)  resident_quads: 256 x 64 for a job that has the chip to itself; a quarter of that when
    // other slots are busy -- the trees then run beside an accumulation kernel that leaves them one wave slot per SIMD at
    // best, and four times the waves for 0.4 x the time cost it 1.5 % (measured, pipelined 2^20: 1024 / 2048 / 4096 / 8192 /
    // 16384 quads gave 374 / 400 / 412 / 412 / 402 commitments/s).
    uint32_t lpg[8];
    if (count > 8) count = 8;
    uint64_t quads = 0;
    for (uint32_t i = 0; i < count; i++) {
        lpg[i] = 1;
        while (lpg[i] < descs[i].len && lpg[i] < kTreeLogical) lpg[i] <<= 1;
        quads += (uint64_t)descs[i].groups * lpg[i];
    }
    const uint64_t resident_lanes = resident_quads;
    while (quads > resident_lanes) {
        uint32_t widest = 1;
        for (uint32_t i = 0; i < count; i++) widest = lpg[i] > widest ? lpg[i] : widest;
        if (widest == 1) break;
        quads = 0;
        for (uint32_t i = 0; i < count; i++) {
            if (lpg[i] == widest) lpg[i] >>= 1;
            quads += (uint64_t)descs[i].groups * lpg[i];
        }
    }
    uint32_t blocks = first_block;
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t gpb = kTreeLogical / lpg[i];
        TreeJob& J = j[i];
        J.in = reinterpret_cast<const uint4*>(descs[i].in);
        J.out = reinterpret_cast<uint4*>(descs[i].out);
        J.groups = descs[i].groups;
        J.len = descs[i].len;
        J.lanes_per_group = lpg[i];
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
