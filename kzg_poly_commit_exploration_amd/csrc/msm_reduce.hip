// msm_reduce.hip -- bucket reduction: sum_b (b+1) * B_b from the 2^(c-1) bucket sums.
//
// The weights are linear in the bucket index, so the weighted sum separates over any split of the
// index into digits: with b = hi * C + lo,
//        sum_b b B_b = C * sum_hi hi * Row_hi + sum_lo lo * Col_lo,
//        Row_hi = sum_lo B[hi*C + lo],   Col_lo = sum_hi B[hi*C + lo],   sum_b B_b = sum_hi Row_hi.
// Rows and columns are PLAIN sums -> logarithmic-depth trees instead of running-sum chains.  The same
// split is applied once more to the Row and the Col vector (<= 1024 entries -> four vectors of <= 32
// entries), whose short weighted sums the host finishes (api.hip).  Critical path on the device:
// log2(C) + log2(sqrt) dependent 384-bit additions (12 at 65536 buckets) instead of ~70 with running sums;
// the work stays 2 additions per bucket.  Everything here is bound by the latency of the field
// multiplier (~27 us per addition for a lone wave), not by HBM or VALU throughput.
//
// Field products stay inlined: measured on MI355X the call-based multiplier costs the general
// addition 40 % (2.8 vs 1.6 G additions/s, gpurun_out microbench), the register traffic around the
// calls outweighing the instruction-cache savings.

// Every addition is spread over the four lanes of a quad (xyzz30_add_quad, g1_30.hip.h; see msm_finalize.hip): a lane
// below is a LOGICAL lane = one quad of the 256-thread workgroup (64 logical lanes).
#define KZG_G1_30_NO_SB 1
#include "engine.h"
#include "g1_30.hip.h"

namespace kzg {

constexpr int kCoop = 4;  // physical lanes per logical lane
#define KZG_TREE_ADD(a, b) xyzz30_add_quad(a, b, threadIdx.x & 3u)
#define KZG_TREE_WAVES 1

constexpr int kTreeBlock = 256;

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
struct TreeJobs {
    TreeJob j[4];
    uint32_t count;
};

__global__ void __launch_bounds__(kTreeBlock, KZG_TREE_WAVES) k_tree_sum(TreeJobs jobs) {
    constexpr int kLogical = kTreeBlock / kCoop;  // logical lanes per workgroup
    __shared__ uint32_t lds[4 * kQ * kLogical];
    const int t = threadIdx.x / kCoop;
    const bool lead = (threadIdx.x & 3u) == 0;
    uint32_t ji = 0;
#pragma unroll
    for (uint32_t q = 1; q < 4; q++)
        if (q < jobs.count && blockIdx.x >= jobs.j[q].first_block) ji = q;
    const TreeJob J = jobs.j[ji];
    const uint32_t lanes_per_group = J.lanes_per_group;
    const uint32_t gpb = kLogical / lanes_per_group;
    const uint32_t g = (blockIdx.x - J.first_block) * gpb + t / lanes_per_group;
    const uint32_t l = t & (lanes_per_group - 1);
    XYZZ30 acc = xyzz30_inf();
    if (g < J.groups) {
        const uint64_t base = (uint64_t)(g / J.inner) * J.ostride + (uint64_t)(g % J.inner) * J.gstride;
        for (uint32_t q = l; q < J.len; q += lanes_per_group) {
            XYZZ30 b = load_xyzz30(J.in + (size_t)(base + q * J.estride) * kXyzzU4);
            KZG_TREE_ADD(acc, b);
        }
    }
    for (uint32_t off = lanes_per_group >> 1; off >= 1; off >>= 1) {
        __syncthreads();
        if (lead && l >= off && l < 2 * off) {
            const Fq* f[4] = {&acc.X, &acc.Y, &acc.ZZ, &acc.ZZZ};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < kQ; i++) lds[(q * kQ + i) * kLogical + (t - off)] = (uint32_t)f[q]->d[i];
        }
        __syncthreads();
        if (l < off) {
            XYZZ30 o;
            Fq* f[4] = {&o.X, &o.Y, &o.ZZ, &o.ZZZ};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < kQ; i++) f[q]->d[i] = (int32_t)lds[(q * kQ + i) * kLogical + t];
            KZG_TREE_ADD(acc, o);
        }
    }
    if (lead && l == 0 && g < J.groups) store_xyzz30(J.out + (size_t)g * kXyzzU4, acc);
}

void launch_tree_sums(hipStream_t s, const TreeSumDesc* descs, uint32_t count) {
    TreeJobs jobs;
    jobs.count = count;
    // One (logical) lane per element gives the shortest chain (log2(len) dependent additions) but only ~1/log2(len)
    // of the lane-steps do work.  A CU holds one workgroup (one wave per SIMD) = 64 logical lanes: beyond 256
    // workgroups a second round would start, which costs a whole tree's latency -- lanes then pre-add several
    // elements serially instead (one more dependent addition per doubling).
    constexpr uint32_t kLogical = kTreeBlock / kCoop;
    uint64_t total = 0;
    for (uint32_t i = 0; i < count && i < 4; i++) total += (uint64_t)descs[i].groups * descs[i].len;
    const uint64_t resident_lanes = (uint64_t)KZG_TREE_WAVES * 256 * kLogical;
    uint32_t per_lane = 1;
    while ((total + per_lane - 1) / per_lane > resident_lanes && per_lane < 64) per_lane <<= 1;
    uint32_t blocks = 0;
    for (uint32_t i = 0; i < count && i < 4; i++) {
        uint32_t lpg = 1;
        while (lpg < descs[i].len && lpg < kLogical) lpg <<= 1;
        lpg = lpg / per_lane ? lpg / per_lane : 1;
        uint32_t gpb = kLogical / lpg;
        jobs.j[i].in = reinterpret_cast<const uint4*>(descs[i].in);
        jobs.j[i].out = reinterpret_cast<uint4*>(descs[i].out);
        jobs.j[i].groups = descs[i].groups;
        jobs.j[i].len = descs[i].len;
        jobs.j[i].lanes_per_group = lpg;
        jobs.j[i].first_block = blocks;
        jobs.j[i].gstride = descs[i].gstride;
        jobs.j[i].estride = descs[i].estride;
        jobs.j[i].inner = descs[i].inner ? descs[i].inner : descs[i].groups;
        jobs.j[i].ostride = descs[i].ostride;
        blocks += (descs[i].groups + gpb - 1) / gpb;
    }
    if (!blocks) return;
    hipLaunchKernelGGL(k_tree_sum, dim3(blocks), dim3(kTreeBlock), 0, s, jobs);
}

}  // namespace kzg
