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
#include "msm_tree_plan.h"

namespace kzg {

constexpr int kCoop = 4;  // physical lanes per logical lane
#define KZG_TREE_ADD(a, b) xyzz30_add_quad(a, b, threadIdx.x & 3u)
#define KZG_TREE_WAVES 1


struct TreeJobs {
    TreeJob j[6];
    uint32_t count;
    // Two dependent stages in ONE launch (small jobs: a launch boundary costs more than a tree level).  sync[0] hands
    // out workgroup numbers in the order the workgroups START, sync[1] counts finished stage-1 workgroups; a workgroup
    // whose number is >= stage1_blocks waits for sync[1] == stage1_blocks.  Whoever holds such a number knows every
    // stage-1 workgroup has started and depends on nothing, so the wait always ends -- whatever order the hardware
    // dispatches workgroups in.  Null: plain launch, workgroup number = blockIdx.x.
    uint32_t stage1_blocks;
    uint32_t* sync;
};

// kDense: idle quads keep their lanes active (xyzz30_add_quad_dense, g1_30.hip.h) -- the latency form, chosen for jobs of
// at most 16384 buckets; above that the trees share the chip with other slots' accumulation kernels and stay sparse.
template <bool kDense>
__global__ void __launch_bounds__(kTreeBlock, KZG_TREE_WAVES) k_tree_sum(TreeJobs jobs) {
    constexpr int kLogical = kTreeBlock / kCoop;  // logical lanes per workgroup
    __shared__ uint32_t lds[4 * kQ * kLogical];
    __shared__ uint32_t s_ticket;
    const int t = threadIdx.x / kCoop;
    const bool lead = (threadIdx.x & 3u) == 0;
    uint32_t bid = blockIdx.x;
    if (jobs.sync) {
        if (threadIdx.x == 0) {
            s_ticket = atomicAdd(&jobs.sync[0], 1u);
            if (s_ticket >= jobs.stage1_blocks)
                while (__hip_atomic_load(&jobs.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < jobs.stage1_blocks)
                    __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
        bid = s_ticket;
        if (bid >= jobs.stage1_blocks) __threadfence();  // acquire: stage-1 results are read below
    }
    uint32_t ji = 0;
#pragma unroll
    for (uint32_t q = 1; q < 6; q++)
        if (q < jobs.count && bid >= jobs.j[q].first_block) ji = q;
    const TreeJob J = jobs.j[ji];
    const uint32_t lanes_per_group = J.lanes_per_group;
    const uint32_t gpb = kLogical / lanes_per_group;
    const uint32_t g = (bid - J.first_block) * gpb + t / lanes_per_group;
    const uint32_t l = t & (lanes_per_group - 1);
    XYZZ30 acc = xyzz30_inf();
    if (g < J.groups) {
        const uint64_t base = (uint64_t)(g / J.inner) * J.ostride + (uint64_t)(g % J.inner) * J.gstride;
        for (uint32_t q = l; q < J.len; q += lanes_per_group) {
            XYZZ30 b = load_xyzz30(J.in + (size_t)(base + q * J.estride) * kXyzzU4);
            if (kDense) xyzz30_add_quad_dense(acc, b, threadIdx.x & 3u);
            else KZG_TREE_ADD(acc, b);
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
        if (kDense) {  // every quad enters; those without a partner bring an operand at infinity
            XYZZ30 o = xyzz30_inf();
            if (l < off) {
                Fq* f[4] = {&o.X, &o.Y, &o.ZZ, &o.ZZZ};
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int i = 0; i < kQ; i++) f[q]->d[i] = (int32_t)lds[(q * kQ + i) * kLogical + t];
            }
            xyzz30_add_quad_dense(acc, o, threadIdx.x & 3u);
        } else if (l < off) {
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
    if (jobs.sync && bid < jobs.stage1_blocks) {
        __threadfence();  // release: this workgroup's results before its count
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&jobs.sync[1], 1u);
    }
}

static uint32_t plan_jobs(TreeJobs& jobs, uint32_t first, const TreeSumDesc* descs, uint32_t count, uint32_t first_block,
                          bool alone = true) {
    return plan_tree_jobs(jobs.j + first, descs, count, first_block, alone ? 256u * kTreeLogical : 64u * kTreeLogical);
}

static bool dense_trees(const TreeSumDesc* descs, uint32_t count) {
    uint64_t total = 0;
    for (uint32_t i = 0; i < count; i++) total += (uint64_t)descs[i].groups * descs[i].len;
    return total <= 2 * 16384;  // (stage 1 sums every bucket twice: rows and columns)
}

void launch_tree_sums(hipStream_t s, const TreeSumDesc* descs, uint32_t count, bool dense, bool alone) {
    if (count > 6) count = 6;
    TreeJobs jobs;
    jobs.count = count;
    jobs.stage1_blocks = 0;
    jobs.sync = nullptr;
    const uint32_t blocks = plan_jobs(jobs, 0, descs, count, 0, alone);
    if (!blocks) return;
    if (dense) hipLaunchKernelGGL(k_tree_sum<true>, dim3(blocks), dim3(kTreeBlock), 0, s, jobs);
    else hipLaunchKernelGGL(k_tree_sum<false>, dim3(blocks), dim3(kTreeBlock), 0, s, jobs);
}

void launch_tree_sums_two_stage(hipStream_t s, const TreeSumDesc* stage1, uint32_t count1, const TreeSumDesc* stage2,
                                uint32_t count2, uint32_t* d_sync, bool alone) {
    TreeJobs jobs;
    jobs.count = count1 + count2;
    const uint32_t b1 = count1 + count2 <= 6 ? plan_jobs(jobs, 0, stage1, count1, 0, alone) : 0;
    const uint32_t b2 = b1 ? plan_jobs(jobs, count1, stage2, count2, b1, alone) : 0;
    // one launch only while every workgroup is resident at once anyway (two per CU); else two launches
    const bool dense = alone || dense_trees(stage1, count1);  // (a lone job has no neighbour whose accumulation the dense form could slow)
    if (!d_sync || !b1 || !b2 || b1 + b2 > 512) {
        launch_tree_sums(s, stage1, count1, dense, alone);
        launch_tree_sums(s, stage2, count2, dense, alone);
        return;
    }
    jobs.stage1_blocks = b1;
    jobs.sync = d_sync;
    if (dense) hipLaunchKernelGGL(k_tree_sum<true>, dim3(b1 + b2), dim3(kTreeBlock), 0, s, jobs);
    else hipLaunchKernelGGL(k_tree_sum<false>, dim3(b1 + b2), dim3(kTreeBlock), 0, s, jobs);
}

}  // namespace kzg
