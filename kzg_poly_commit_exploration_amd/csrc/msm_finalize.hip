// msm_finalize.hip -- completes the buckets whose references were split over several accumulation
// segments (msm_accum.hip): adds the head / tail partials.  Kept in its own translation unit so that
// its register allocation cannot perturb the dominant kernel's.
//
// A bucket of the sorted reference list [s, e) leaves one partial per segment it touches ("pieces": the tail
// of the first segment, whole middle segments, the head of the last one).  Few pieces (<= kSerialSpan, every
// bucket of a uniform input) are added by the bucket's own lane.  Long runs -- skewed inputs: a 0/1
// polynomial puts all 2^20 references into ONE bucket, small i128 coefficients into a few thousand -- are
// reduced by key with 64-wide LDS trees in ONE more launch: every workgroup folds a chunk of 64 pieces; the
// workgroup that finishes the last chunk of a group of 64 chunks folds their results, and the one that
// finishes an entry's last group folds the groups (64^3 = 262144 >= the number of segments).  The critical
// path is 3 x 6 dependent additions instead of one per piece.  Bound by the latency of the field multiplier
// (~27 us per dependent addition).  One launch, not three: every launch of a kernel with ~180 VGPRs has to
// wait for register space next to the other slots' accumulation kernels, whether or not it finds work.
// Inline doubling in the (rare) equal-operands branch: keeps these kernels free of scratch memory.

// Every addition here is spread over the four lanes of a quad (xyzz30_add_quad, g1_30.hip.h): these kernels are bound
// by chains of dependent additions, and four stages of one product each finish a general addition 2.5 x sooner than
// one lane's fourteen products in a row.  A "lane" below is therefore a LOGICAL lane = one quad; q = threadIdx.x & 3
// is the position inside it; loads are issued by all four lanes (same address: one request), stores, atomics and LDS
// writes by lane 0 of the quad.  No scheduling barriers between the products (one wave per SIMD, registers to spare).
#define KZG_G1_30_NO_SB 1
#include "engine.h"
#include "g1_30.hip.h"

namespace kzg {

constexpr int kCoop = 4;  // physical lanes per logical lane
#define KZG_TREE_ADD(a, b) xyzz30_add_quad(a, b, threadIdx.x & 3u)
#define KZG_TREE_WAVES 1

constexpr uint32_t kSerialSpan = 16;  // buckets spanning more segments than this go through the tree kernel
constexpr uint32_t kSerialSpanFew = 4, kFewBuckets = 2048;  // threshold when there are at most kFewBuckets buckets
constexpr int kChunk = 64;           // pieces per tree = lanes per workgroup of the tree passes
constexpr int kTreeGrid = 256;       // workgroups per tree pass (grid-stride over the work items): one resident round (256 lanes = one wave per SIMD)

// one registered long bucket
struct HeavyEntry {
    uint32_t bucket, l_lo, span, first_is_b;  // pieces: segments l_lo .. l_lo+span-1; first piece from part_b?
    uint32_t base1, c1;                        // its c1 = ceil(span/64) chunks: results in tmp1[base1 ..]
    uint32_t base2, c2;                        // its c2 = ceil(c1/64) groups of chunks: results in tmp2[base2 ..]
};
static_assert(sizeof(HeavyEntry) == 32, "layout used by heavy_workspace_bytes");

// workspace layout: counters[8] | entries | entry_done | owner1 | group_done | tmp1 (XYZZ) | tmp2 (XYZZ)
// Bounds: a registered bucket has >= kSerialSpan + 1 pieces, of which only the first and the last segment can be
// shared with a neighbour, so it owns >= kSerialSpan - 1 segments outright; sum of spans <= segments + entries.
constexpr size_t kMaxEntries = kMaxAccumLanes / (kSerialSpan - 1) + 1;
constexpr size_t kMaxChunks1 = (kMaxAccumLanes + kMaxEntries) / kChunk + kMaxEntries + 2;  // sum ceil(span/64)
constexpr size_t kMaxChunks2 = kMaxChunks1 / kChunk + kMaxEntries + 2;                     // sum ceil(c1/64)
struct HeavyWs {
    uint32_t* counters;    // [0] entries, [1] chunks, [2] groups
    HeavyEntry* entries;
    uint32_t* entry_done;  // groups finished, per entry
    uint32_t* owner1;      // chunk -> entry
    uint32_t* group_done;  // chunks finished, per group
    uint4 *tmp1, *tmp2;
};
size_t heavy_workspace_bytes() {
    return 32 + kMaxEntries * (sizeof(HeavyEntry) + 4) + (kMaxChunks1 + kMaxChunks2) * 4 + 64 +
           (kMaxChunks1 + kMaxChunks2) * kXyzzBytes;
}
static HeavyWs carve(void* base) {
    char* p = (char*)base;
    HeavyWs w;
    w.counters = (uint32_t*)p; p += 32;
    w.entries = (HeavyEntry*)p; p += kMaxEntries * sizeof(HeavyEntry);
    w.entry_done = (uint32_t*)p; p += kMaxEntries * 4;
    w.owner1 = (uint32_t*)p; p += kMaxChunks1 * 4;
    w.group_done = (uint32_t*)p; p += kMaxChunks2 * 4;
    p = (char*)(((uintptr_t)p + 63) & ~(uintptr_t)63);
    w.tmp1 = (uint4*)p; p += kMaxChunks1 * kXyzzBytes;
    w.tmp2 = (uint4*)p;
    return w;
}

// One lane per bucket: short runs are added here, long ones registered for the tree passes.
__global__ void __launch_bounds__(64 * kCoop, KZG_TREE_WAVES) k_bucket_finalize(const uint32_t* __restrict__ offs, uint32_t nb, uint32_t lanes,
                                                        const uint4* __restrict__ part_a,
                                                        const uint4* __restrict__ part_b,
                                                        uint4* __restrict__ buckets, HeavyWs ws,
                                                        uint32_t* __restrict__ refs_out) {
    const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) / kCoop;  // one quad per bucket
    const bool lead = (threadIdx.x & 3u) == 0;
    if (b == 0 && lead && refs_out) refs_out[0] = offs[nb];  // number of references, for the host's statistics
    if (b >= nb) return;
    const uint32_t L = accumulate_seg_len(offs[nb], lanes);
    uint32_t s = offs[b], e = offs[b + 1];
    if (s == e) return;  // empty bucket: stays at infinity (buffer pre-zeroed)
    uint32_t l_lo = s / L, l_hi = (e - 1) / L;
    if (l_lo == l_hi) return;  // inside one segment: written complete by k_bucket_accumulate
    const uint32_t span = l_hi - l_lo + 1;
    const bool first_is_b = s != l_lo * L;
    // few buckets (small commitments, latency-bound): a 64-wide tree of 14 pieces is 4 dependent additions, the serial
    // loop 13; with many buckets the serial loops run side by side and the tree kernel would need several rounds
    const uint32_t serial_span = nb <= kFewBuckets ? kSerialSpanFew : kSerialSpan;
    if (span > serial_span) {
        if (!lead) return;
        HeavyEntry en;
        en.bucket = b; en.l_lo = l_lo; en.span = span; en.first_is_b = first_is_b ? 1u : 0u;
        en.c1 = (span + kChunk - 1) / kChunk;
        en.base1 = atomicAdd(&ws.counters[1], en.c1);
        en.c2 = (en.c1 + kChunk - 1) / kChunk;
        en.base2 = atomicAdd(&ws.counters[2], en.c2);
        const uint32_t slot = atomicAdd(&ws.counters[0], 1u);
        ws.entries[slot] = en;
        ws.entry_done[slot] = 0;
        for (uint32_t j = 0; j < en.c1; j++) ws.owner1[en.base1 + j] = slot;
        for (uint32_t j = 0; j < en.c2; j++) ws.group_done[en.base2 + j] = 0;
        return;
    }
    const uint4* first = first_is_b ? part_b + (size_t)l_lo * kXyzzU4 : part_a + (size_t)l_lo * kXyzzU4;
    XYZZ30 acc = load_xyzz30(first);
    for (uint32_t l = l_lo + 1; l <= l_hi; l++) {
        XYZZ30 p = load_xyzz30(part_a + (size_t)l * kXyzzU4);
        KZG_TREE_ADD(acc, p);
    }
    if (lead) store_xyzz30(buckets + (size_t)b * kXyzzU4, acc);
}

// sum of the `count` (<= 64, workgroup-uniform) accumulators held by lanes 0..count-1; result in lane 0
__device__ __forceinline__ void tree64(XYZZ30& acc, uint32_t count, uint32_t* lds /* 52 * 64 words */) {
    const int t = threadIdx.x / kCoop;  // logical lane
    const bool lead = (threadIdx.x & 3u) == 0;
    int top = 1;
    while (top < (int)count && top < kChunk) top <<= 1;
    for (int off = top >> 1; off >= 1; off >>= 1) {
        __syncthreads();
        if (lead && t >= off && t < 2 * off) {
            const Fq* f[4] = {&acc.X, &acc.Y, &acc.ZZ, &acc.ZZZ};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < kQ; i++) lds[(q * kQ + i) * kChunk + (t - off)] = (uint32_t)f[q]->d[i];
        }
        __syncthreads();
        if (t < off) {
            XYZZ30 o;
            Fq* f[4] = {&o.X, &o.Y, &o.ZZ, &o.ZZZ};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < kQ; i++) f[q]->d[i] = (int32_t)lds[(q * kQ + i) * kChunk + t];
            KZG_TREE_ADD(acc, o);
        }
    }
}

// Work items: the chunks of 64 pieces of every registered bucket.  The last workgroup to finish a chunk of a
// group folds the group, the last to finish a group of an entry folds the entry (release: result stored, fence,
// counter incremented; acquire: counter seen complete, fence, results loaded).
__global__ void __launch_bounds__(kChunk * kCoop, KZG_TREE_WAVES) k_heavy_tree(const uint4* __restrict__ part_a,
                                                       const uint4* __restrict__ part_b,
                                                       uint4* __restrict__ buckets, HeavyWs ws) {
    __shared__ uint32_t lds[4 * kQ * kChunk];
    __shared__ uint32_t s_last;
    const uint32_t t = threadIdx.x / kCoop;             // logical lane
    const bool lead = (threadIdx.x & 3u) == 0;
    const uint32_t total = ws.counters[1];
    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
        __syncthreads();  // s_last / lds of the previous item are no longer read
        const uint32_t slot = ws.owner1[item];
        const HeavyEntry en = ws.entries[slot];
        uint4* const bucket = buckets + (size_t)en.bucket * kXyzzU4;
        // level 1: 64 pieces
        const uint32_t j = item - en.base1;
        const uint32_t first = j * kChunk;
        uint32_t count = en.span - first < (uint32_t)kChunk ? en.span - first : (uint32_t)kChunk;
        XYZZ30 acc = xyzz30_inf();
        if (t < count) {
            const uint32_t l = en.l_lo + first + t;
            const uint4* src = (l == en.l_lo && en.first_is_b) ? part_b + (size_t)l * kXyzzU4 : part_a + (size_t)l * kXyzzU4;
            acc = load_xyzz30(src);
        }
        tree64(acc, count, lds);
        if (en.c1 == 1) {
            if (t == 0 && lead) store_xyzz30(bucket, acc);
            continue;
        }
        // level 2: the chunk results of group g, by whoever completes it
        const uint32_t g = j / kChunk;
        const uint32_t in_group = en.c1 - g * kChunk < (uint32_t)kChunk ? en.c1 - g * kChunk : (uint32_t)kChunk;
        if (t == 0 && lead) {
            store_xyzz30(ws.tmp1 + (size_t)item * kXyzzU4, acc);
            __threadfence();
            s_last = atomicAdd(&ws.group_done[en.base2 + g], 1u) == in_group - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (!s_last) continue;
        __threadfence();
        acc = xyzz30_inf();
        if (t < in_group) acc = load_xyzz30(ws.tmp1 + (size_t)(en.base1 + g * kChunk + t) * kXyzzU4);
        tree64(acc, in_group, lds);
        if (en.c2 == 1) {
            if (t == 0 && lead) store_xyzz30(bucket, acc);
            continue;
        }
        // level 3: the group results of the entry
        __syncthreads();
        if (t == 0 && lead) {
            store_xyzz30(ws.tmp2 + (size_t)(en.base2 + g) * kXyzzU4, acc);
            __threadfence();
            s_last = atomicAdd(&ws.entry_done[slot], 1u) == en.c2 - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (!s_last) continue;
        __threadfence();
        acc = xyzz30_inf();
        for (uint32_t i = t; i < en.c2; i += kChunk) {  // c2 <= 64 for <= 262144 segments; strided for safety
            XYZZ30 p = load_xyzz30(ws.tmp2 + (size_t)(en.base2 + i) * kXyzzU4);
            KZG_TREE_ADD(acc, p);
        }
        tree64(acc, en.c2 < (uint32_t)kChunk ? en.c2 : (uint32_t)kChunk, lds);
        if (t == 0 && lead) store_xyzz30(bucket, acc);
    }
}

void launch_bucket_finalize(hipStream_t s, const uint32_t* d_offs, uint32_t nb, uint32_t lanes, const void* d_part_a,
                            const void* d_part_b, void* d_buckets, void* d_heavy_ws, uint32_t* d_refs_out) {
    HeavyWs ws = carve(d_heavy_ws);
    const uint4* pa = reinterpret_cast<const uint4*>(d_part_a);
    const uint4* pb = reinterpret_cast<const uint4*>(d_part_b);
    uint4* bk = reinterpret_cast<uint4*>(d_buckets);
    hipLaunchKernelGGL(k_bucket_finalize, dim3((nb + 63) / 64), dim3(64 * kCoop), 0, s, d_offs, nb, lanes, pa, pb, bk, ws, d_refs_out);
    hipLaunchKernelGGL(k_heavy_tree, dim3(kTreeGrid), dim3(kChunk * kCoop), 0, s, pa, pb, bk, ws);
}

}  // namespace kzg
