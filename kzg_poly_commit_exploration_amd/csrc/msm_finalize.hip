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
#include "msm_tree_plan.h"

namespace kzg {

constexpr int kCoop = 4;  // physical lanes per logical lane
#define KZG_TREE_ADD(a, b) xyzz30_add_quad(a, b, threadIdx.x & 3u)
// (latency form: every lane of the wave stays active, g1_30.hip.h; costs the other slots' kernels SIMD time, so the
// throughput kernels above 65536 references keep the sparse form)
#define KZG_TREE_ADD_DENSE(a, b) xyzz30_add_quad_dense(a, b, threadIdx.x & 3u)
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

// workspace layout: header (kHeavyHeaderBytes: counters[0..2], phase counters) | entries | entry_done | owner1 | group_done | tmp1 (XYZZ) | tmp2 (XYZZ)
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
    return kHeavyHeaderBytes + kMaxEntries * (sizeof(HeavyEntry) + 4) + (kMaxChunks1 + kMaxChunks2) * 4 + 64 +
           (kMaxChunks1 + kMaxChunks2) * kXyzzBytes;
}
static HeavyWs carve(void* base) {
    char* p = (char*)base;
    HeavyWs w;
    w.counters = (uint32_t*)p; p += kHeavyHeaderBytes;
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
    if (b == 0 && lead && refs_out) {
        refs_out[0] = offs[nb];  // number of references, for the host's statistics
#pragma unroll
        for (int i = 0; i < 4; i++) refs_out[2 + i] = ws.counters[kAccumClockOffset / 4 + i];  // the accumulation's start / end stamps
    }
    if (b >= nb) return;
    const uint32_t L = accumulate_seg_len(offs[nb], lanes);
    uint32_t s = offs[b], e = offs[b + 1];
    if (s == e) {  // empty bucket: written as infinity here (the bucket array is not cleared between jobs)
        if (lead) store_xyzz30(buckets + (size_t)b * kXyzzU4, xyzz30_inf());
        return;
    }
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

// ---- a group of quads per bucket -------------------------------------------------------------------------------------
// With few buckets (up to 16384: commitments of up to ~2^15 terms) the serial loop of k_bucket_finalize is the
// longest chain of the job.  k_bucket_finalize_group gives every bucket `group` quads (a power of two <= 16, workgroup
// = 64 quads): each adds every group-th piece, then the group folds through LDS -- pieces/group + log2(group) dependent
// additions instead of one per piece.  Buckets beyond 8 pieces per quad still go to the long-bucket trees.
// Small jobs (<= 65536 references) run the same steps as phases of ONE launch: k_small_msm below.
__device__ __forceinline__ uint32_t bucket_of_pos(const uint32_t* __restrict__ offs, uint32_t nb, uint32_t pos) {
    uint32_t lo = 0, hi = nb;  // invariant: offs[lo] <= pos, (hi == nb or offs[hi] > pos)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offs[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ XYZZ30 load_table_point(const uint4* __restrict__ table, uint32_t ref) {
    const uint4* src = table + (size_t)(ref & 0x7fffffffu) * kAffineU4;
    XYZZ30 r;
    r.X = load_fq16(src);
    r.Y = load_fq16(src + 4);
    if (fq_all_zero(r.X) && fq_all_zero(r.Y)) return xyzz30_inf();  // the table's encoding of infinity
    if (ref >> 31) r.Y = fq_neg(r.Y);
    r.ZZ = fq_one();
    r.ZZZ = r.ZZ;
    return r;
}
constexpr uint32_t kGroupSerial = 8;  // pieces per quad before a bucket goes to the long-bucket trees instead ...
// ... but never fewer than one chunk of those trees in all: at 2^16 terms (4096 buckets of ~32 pieces, four quads each)
// a limit of 32 sent every second bucket there (235 us)
__host__ __device__ inline uint32_t group_span_limit(uint32_t group) {
    return kGroupSerial * group > (uint32_t)kChunk ? kGroupSerial * group : (uint32_t)kChunk;
}
__global__ void __launch_bounds__(64 * kCoop, KZG_TREE_WAVES) k_bucket_finalize_group(
    const uint32_t* __restrict__ offs, uint32_t nb, uint32_t lanes, const uint4* __restrict__ part_a,
    const uint4* __restrict__ part_b, uint4* __restrict__ buckets, HeavyWs ws, uint32_t* __restrict__ refs_out,
    uint32_t group) {
    __shared__ uint32_t lds[4 * kQ * 64];
    const uint32_t t = threadIdx.x / kCoop;  // logical lane 0..63
    const bool lead = (threadIdx.x & 3u) == 0;
    const uint32_t b = blockIdx.x * (64 / group) + t / group;
    const uint32_t l = t & (group - 1);
    if (blockIdx.x == 0 && threadIdx.x == 0 && refs_out) {
        refs_out[0] = offs[nb];
#pragma unroll
        for (int i = 0; i < 4; i++) refs_out[2 + i] = ws.counters[kAccumClockOffset / 4 + i];
    }
    // every thread reaches the barriers below: inactive groups carry infinity through them
    uint32_t span = 0, l_lo = 0;
    bool first_is_b = false;
    if (b < nb) {
        const uint32_t L = accumulate_seg_len(offs[nb], lanes);
        const uint32_t s = offs[b], e = offs[b + 1];
        if (s == e) {  // empty bucket: written as infinity here (the bucket array is not cleared between jobs)
            if (l == 0 && lead) store_xyzz30(buckets + (size_t)b * kXyzzU4, xyzz30_inf());
        } else {
            l_lo = s / L;
            const uint32_t l_hi = (e - 1) / L;
            if (l_lo != l_hi) {  // (inside one segment: written complete by the accumulation kernel)
                span = l_hi - l_lo + 1;
                first_is_b = s != l_lo * L;
                if (span > group_span_limit(group)) {
                    if (l == 0 && lead) {
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
                    }
                    span = 0;
                }
            }
        }
    }
    XYZZ30 acc = xyzz30_inf();
    for (uint32_t i = l; i < span; i += group) {
        const uint4* src = (i == 0 && first_is_b) ? part_b + (size_t)l_lo * kXyzzU4 : part_a + (size_t)(l_lo + i) * kXyzzU4;
        const XYZZ30 p = load_xyzz30(src);
        KZG_TREE_ADD_DENSE(acc, p);
    }
    for (uint32_t off = group >> 1; off >= 1; off >>= 1) {
        __syncthreads();
        if (lead && l >= off && l < 2 * off) {
            const Fq* f[4] = {&acc.X, &acc.Y, &acc.ZZ, &acc.ZZZ};
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int i = 0; i < kQ; i++) lds[(c * kQ + i) * 64 + (t - off)] = (uint32_t)f[c]->d[i];
        }
        __syncthreads();
        if (l < off) {
            XYZZ30 o;
            Fq* f[4] = {&o.X, &o.Y, &o.ZZ, &o.ZZZ};
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int i = 0; i < kQ; i++) f[c]->d[i] = (int32_t)lds[(c * kQ + i) * 64 + t];
            KZG_TREE_ADD_DENSE(acc, o);
        }
    }
    if (span && l == 0 && lead) store_xyzz30(buckets + (size_t)b * kXyzzU4, acc);
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
// one work item of the long-bucket trees (a chunk of 64 pieces and whatever its completion completes)
static __device__ void heavy_item(uint32_t item, const uint4* __restrict__ part_a, const uint4* __restrict__ part_b,
                                  uint4* __restrict__ buckets, const HeavyWs& ws, uint32_t* lds, uint32_t* s_last) {
    const uint32_t t = threadIdx.x / kCoop;  // logical lane
    const bool lead = (threadIdx.x & 3u) == 0;
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
        return;
    }
    // level 2: the chunk results of group g, by whoever completes it
    const uint32_t g = j / kChunk;
    const uint32_t in_group = en.c1 - g * kChunk < (uint32_t)kChunk ? en.c1 - g * kChunk : (uint32_t)kChunk;
    if (t == 0 && lead) {
        store_xyzz30(ws.tmp1 + (size_t)item * kXyzzU4, acc);
        __threadfence();
        *s_last = atomicAdd(&ws.group_done[en.base2 + g], 1u) == in_group - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!*s_last) return;
    __threadfence();
    acc = xyzz30_inf();
    if (t < in_group) acc = load_xyzz30(ws.tmp1 + (size_t)(en.base1 + g * kChunk + t) * kXyzzU4);
    tree64(acc, in_group, lds);
    if (en.c2 == 1) {
        if (t == 0 && lead) store_xyzz30(bucket, acc);
        return;
    }
    // level 3: the group results of the entry
    __syncthreads();
    if (t == 0 && lead) {
        store_xyzz30(ws.tmp2 + (size_t)(en.base2 + g) * kXyzzU4, acc);
        __threadfence();
        *s_last = atomicAdd(&ws.entry_done[slot], 1u) == en.c2 - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!*s_last) return;
    __threadfence();
    acc = xyzz30_inf();
    for (uint32_t i = t; i < en.c2; i += kChunk) {  // c2 <= 64 for <= 262144 segments; strided for safety
        XYZZ30 p = load_xyzz30(ws.tmp2 + (size_t)(en.base2 + i) * kXyzzU4);
        KZG_TREE_ADD(acc, p);
    }
    tree64(acc, en.c2 < (uint32_t)kChunk ? en.c2 : (uint32_t)kChunk, lds);
    if (t == 0 && lead) store_xyzz30(bucket, acc);
}

__global__ void __launch_bounds__(kChunk * kCoop, KZG_TREE_WAVES) k_heavy_tree(const uint4* __restrict__ part_a,
                                                       const uint4* __restrict__ part_b,
                                                       uint4* __restrict__ buckets, HeavyWs ws) {
    __shared__ uint32_t lds[4 * kQ * kChunk];
    __shared__ uint32_t s_last;
    const uint32_t total = ws.counters[1];
    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
        __syncthreads();  // s_last / lds of the previous item are no longer read
        heavy_item(item, part_a, part_b, buckets, ws, lds, &s_last);
    }
}

// ---- small jobs in ONE launch ---------------------------------------------------------------------------------------
// Everything between the sort and the copy back to the host -- accumulation, bucket finalisation, long-bucket trees, the
// two stages of the Row / Col reduction -- as work items of one persistent kernel.  Why: at a few thousand terms each
// of those kernels runs a handful of dependent additions, and a launch boundary (5-12 us with its stream hand-over),
// the instruction fetch of a fresh 40 KB copy of the addition per kernel and per call site (~7 us each, measured
// with in-kernel timestamps) and two workgroups landing on one CU (every addition twice as slow) cost more than
// the arithmetic.  Here there is ONE call site of the addition (the step loop below), one workgroup per CU (the
// launch reserves more than half of a CU's LDS), and the phases are separated by counters instead of launches:
//   counters[32]  ticket: work items are numbered in phase order and handed out in the order workgroups ask;
//   counters[64], [96], [128], [160]  finished items of phase A (accumulate), B (finalise), H (long buckets), C + D
//   (reduction stages) -- one 128-byte line each: the waiting workgroups poll one line, the tickets go through another.
// A workgroup holding an item of a later phase waits until the earlier phase has finished all its items.  Every
// earlier item has been handed out by then (tickets are in phase order) to a workgroup that is running and depends
// only on phases before its own, so the wait ends whatever the number of resident workgroups and whatever order the
// hardware starts them in.  The number of long-bucket items is only known once phase B is complete (counters[1]).
struct SmallJob {
    const uint4* table;
    const uint32_t* sorted;
    const uint32_t* offs;
    uint4 *buckets, *part_a, *part_b;
    uint32_t* refs_out;
    HeavyWs ws;
    TreeJob tj[6];                    // [0..1] stage 1 (Row, Col), [2..5] stage 2
    uint32_t nb, lanes, group;        // accumulation / finalisation geometry
    uint32_t direct;                  // no phase A: the bucket groups of phase B add table points themselves
    uint32_t items_a, items_b, items_c, items_d;
};

constexpr uint32_t kDirectGroupRefs = 128;  // direct mode: references of one bucket a group of quads adds by itself
constexpr uint32_t kSyncTicket = 32, kSyncDoneA = 64, kSyncDoneB = 96, kSyncDoneH = 128, kSyncDoneC = 160;  // words of the header
__device__ __forceinline__ void spin_until(const uint32_t* p, uint32_t n) {
    // ~0.4 us between polls: up to 255 workgroups wait on one word that the working ones have to increment
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) __builtin_amdgcn_s_sleep(16);
}

__global__ void __launch_bounds__(64 * kCoop, 1) k_small_msm(SmallJob job) {
    __shared__ uint32_t lds[4 * kQ * 64];
    __shared__ uint32_t s_ticket, s_last, s_max;
    const uint32_t t = threadIdx.x / kCoop;  // logical lane (quad) 0..63
    const bool lead = (threadIdx.x & 3u) == 0;
    uint32_t* const sync = job.ws.counters;
    const uint32_t IA = job.items_a, IB = job.items_b, IC = job.items_c, ID = job.items_d;
    for (;;) {
        __syncthreads();  // the previous item no longer reads lds / s_ticket
        if (threadIdx.x == 0) {
            const uint32_t T = atomicAdd(&sync[kSyncTicket], 1u);
            uint32_t phase = 5, item = 0;
            if (T < IA) {
                phase = 0, item = T;
            } else if (T < IA + IB) {
                spin_until(&sync[kSyncDoneA], IA);
                phase = 1, item = T - IA;
            } else {
                spin_until(&sync[kSyncDoneB], IB);
                const uint32_t nH = __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t T2 = T - IA - IB;
                if (T2 < nH) {
                    phase = 2, item = T2;
                } else if (T2 - nH < IC) {
                    spin_until(&sync[kSyncDoneH], nH);
                    phase = 3, item = T2 - nH;
                } else if (T2 - nH < IC + ID) {
                    spin_until(&sync[kSyncDoneC], IC);
                    phase = 4, item = T2 - nH - IC;
                }
            }
            s_ticket = phase | (item << 3);
            s_max = 0;
        }
        __syncthreads();
        const uint32_t phase = s_ticket & 7u, item = s_ticket >> 3;
        if (phase == 5) break;
        if (phase) __threadfence();  // acquire: results of the earlier phases are read below
        if (phase == 2 && !job.direct) {
            heavy_item(item, job.part_a, job.part_b, job.buckets, job.ws, lds, &s_last);
        } else {
            // ---- set up this quad's part of the item ----
            XYZZ30 acc = xyzz30_inf(), next = xyzz30_inf();
            // phase A: one segment of the sorted references
            uint32_t start = 0, end = 0, b = 0, b_beg = 0, b_end = 0, run_start = 0, lane = 0;
            // phases B, C, D: `cnt` operands at p_base + i * p_stride (the first one at p_first), then a tree over
            // the gsz quads of the group; result to `out` (quad 0 of the group) or nowhere
            const uint4* p_first = nullptr;
            const uint4* p_base = nullptr;
            size_t p_stride = 0;
            uint32_t cnt = 0, gsz = 1, l = 0;
            uint4* out = nullptr;
            // direct mode (phases B and H): the operands are table points, reference i of this quad at r_base[i * gsz]
            const uint32_t* r_base = nullptr;
            if (phase == 0) {
                if (item == 0 && threadIdx.x == 0 && job.refs_out) job.refs_out[0] = job.offs[job.nb];
                lane = item * 64 + t;
                const uint32_t M = job.offs[job.nb];
                const uint32_t L = accumulate_seg_len(M, job.lanes);
                const uint64_t start64 = (uint64_t)lane * L;
                if (start64 < M) {
                    start = (uint32_t)start64;
                    end = (M - start < L) ? M : start + L;
                    b = bucket_of_pos(job.offs, job.nb, start);
                    b_beg = job.offs[b];
                    b_end = job.offs[b + 1];
                    run_start = start;
                    next = load_table_point(job.table, job.sorted[start]);
                }
                cnt = end - start;
            } else if (job.direct && phase <= 2) {
                // phase B: a group of quads per bucket; a bucket with more than kDirectGroupRefs references is left to a
                // whole workgroup of phase H (listed in ws.owner1, counted in counters[1])
                if (phase == 1 && item == 0 && threadIdx.x == 0 && job.refs_out) job.refs_out[0] = job.offs[job.nb];
                gsz = phase == 1 ? job.group : 64u;
                l = t & (gsz - 1);
                const uint32_t bk = phase == 1 ? item * (64 / gsz) + t / gsz : job.ws.owner1[item];
                if (bk < job.nb) {
                    const uint32_t s = job.offs[bk], e = job.offs[bk + 1];
                    if (s == e) {
                        if (l == 0 && lead) store_xyzz30(job.buckets + (size_t)bk * kXyzzU4, xyzz30_inf());
                    } else if (phase == 1 && e - s > kDirectGroupRefs) {
                        if (l == 0 && lead) job.ws.owner1[atomicAdd(&job.ws.counters[1], 1u)] = bk;
                    } else {
                        cnt = e - s > l ? (e - s - l + gsz - 1) / gsz : 0;
                        r_base = job.sorted + s + l;
                        if (cnt) next = load_table_point(job.table, r_base[0]);
                        if (l == 0) out = job.buckets + (size_t)bk * kXyzzU4;
                    }
                }
            } else if (phase == 1) {
                gsz = job.group;
                l = t & (gsz - 1);
                const uint32_t bk = item * (64 / gsz) + t / gsz;
                if (bk < job.nb) {
                    const uint32_t L = accumulate_seg_len(job.offs[job.nb], job.lanes);
                    const uint32_t s = job.offs[bk], e = job.offs[bk + 1];
                    if (s == e) {
                        if (l == 0 && lead) store_xyzz30(job.buckets + (size_t)bk * kXyzzU4, xyzz30_inf());
                    } else {
                        const uint32_t l_lo = s / L, l_hi = (e - 1) / L;
                        if (l_lo != l_hi) {  // (inside one segment: written complete by phase A)
                            const uint32_t span = l_hi - l_lo + 1;
                            const bool first_is_b = s != l_lo * L;
                            if (span > group_span_limit(gsz)) {
                                if (l == 0 && lead) {
                                    HeavyEntry en;
                                    en.bucket = bk; en.l_lo = l_lo; en.span = span; en.first_is_b = first_is_b ? 1u : 0u;
                                    en.c1 = (span + kChunk - 1) / kChunk;
                                    en.base1 = atomicAdd(&job.ws.counters[1], en.c1);
                                    en.c2 = (en.c1 + kChunk - 1) / kChunk;
                                    en.base2 = atomicAdd(&job.ws.counters[2], en.c2);
                                    const uint32_t slot = atomicAdd(&job.ws.counters[0], 1u);
                                    job.ws.entries[slot] = en;
                                    job.ws.entry_done[slot] = 0;
                                    for (uint32_t j = 0; j < en.c1; j++) job.ws.owner1[en.base1 + j] = slot;
                                    for (uint32_t j = 0; j < en.c2; j++) job.ws.group_done[en.base2 + j] = 0;
                                }
                            } else {
                                cnt = span > l ? (span - l + gsz - 1) / gsz : 0;
                                p_base = job.part_a + (size_t)(l_lo + l) * kXyzzU4;
                                p_first = (l == 0 && first_is_b) ? job.part_b + (size_t)l_lo * kXyzzU4 : p_base;
                                p_stride = (size_t)gsz * kXyzzU4;
                                if (l == 0) out = job.buckets + (size_t)bk * kXyzzU4;
                            }
                        }
                    }
                }
            } else {
                uint32_t ji = phase == 3 ? 0u : 2u;
                const uint32_t last = phase == 3 ? 1u : 5u;
                const uint32_t blk = item + (phase == 3 ? 0u : job.tj[2].first_block);
                for (uint32_t k = ji + 1; k <= last; k++)
                    if (blk >= job.tj[k].first_block) ji = k;
                const TreeJob& J = job.tj[ji];
                gsz = J.lanes_per_group;
                l = t & (gsz - 1);
                const uint32_t g = (blk - J.first_block) * (64 / gsz) + t / gsz;
                if (g < J.groups) {
                    const uint64_t base = (uint64_t)(g / J.inner) * J.ostride + (uint64_t)(g % J.inner) * J.gstride;
                    cnt = J.len > l ? (J.len - l + gsz - 1) / gsz : 0;
                    p_base = J.in + (size_t)(base + l * J.estride) * kXyzzU4;
                    p_first = p_base;
                    p_stride = (size_t)gsz * J.estride * kXyzzU4;
                    if (l == 0) out = J.out + (size_t)g * kXyzzU4;
                }
            }
            if (phase != 0 && !r_base && cnt) next = load_xyzz30(p_first);
            atomicMax(&s_max, cnt);
            __syncthreads();
            const uint32_t n_serial = s_max;
            uint32_t levels = 0;
            while ((1u << levels) < gsz) levels++;
            // ---- the step loop: the one call site of the addition ----
            for (uint32_t step = 0; step < n_serial + levels; step++) {
                XYZZ30 o = xyzz30_inf();
                if (step < n_serial) {
                    if (step < cnt) {
                        o = next;
                        if (phase == 0) {
                            const uint32_t e = start + step;
                            if (e == b_end) {  // bucket b ends here: flush its run, move on to the bucket that owns e
                                uint4* dst = (run_start == b_beg) ? job.buckets + (size_t)b * kXyzzU4 : job.part_a + (size_t)lane * kXyzzU4;
                                if (lead) store_xyzz30(dst, acc);
                                acc = xyzz30_inf();
                                do {
                                    b++;
                                    b_beg = b_end;
                                    b_end = job.offs[b + 1];
                                } while (b_end <= e);
                                run_start = e;
                            }
                            if (step + 1 < cnt) next = load_table_point(job.table, job.sorted[e + 1]);
                        } else if (step + 1 < cnt) {
                            next = r_base ? load_table_point(job.table, r_base[(size_t)(step + 1) * gsz])
                                          : load_xyzz30(p_base + (size_t)(step + 1) * p_stride);
                        }
                    }
                } else {
                    const uint32_t off = gsz >> (step - n_serial + 1);
                    __syncthreads();
                    if (lead && l >= off && l < 2 * off) {
                        const Fq* f[4] = {&acc.X, &acc.Y, &acc.ZZ, &acc.ZZZ};
#pragma unroll
                        for (int c = 0; c < 4; c++)
#pragma unroll
                            for (int i = 0; i < kQ; i++) lds[(c * kQ + i) * 64 + (t - off)] = (uint32_t)f[c]->d[i];
                    }
                    __syncthreads();
                    if (l < off) {
                        Fq* f[4] = {&o.X, &o.Y, &o.ZZ, &o.ZZZ};
#pragma unroll
                        for (int c = 0; c < 4; c++)
#pragma unroll
                            for (int i = 0; i < kQ; i++) f[c]->d[i] = (int32_t)lds[(c * kQ + i) * 64 + t];
                    }
                }
                KZG_TREE_ADD_DENSE(acc, o);
            }
            if (phase == 0) {
                if (cnt) {  // last run: [run_start, end)
                    uint4* dst;
                    if (run_start == b_beg && end == b_end) dst = job.buckets + (size_t)b * kXyzzU4;  // complete
                    else if (run_start == start) dst = job.part_a + (size_t)lane * kXyzzU4;            // covers the whole segment
                    else dst = job.part_b + (size_t)lane * kXyzzU4;                                    // tail shared with the next lane
                    if (lead) store_xyzz30(dst, acc);
                }
            } else if (out && lead) {
                store_xyzz30(out, acc);
            }
        }
        // release: this item's results before its count
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&sync[phase == 0 ? kSyncDoneA : phase == 1 ? kSyncDoneB : phase == 2 ? kSyncDoneH : kSyncDoneC], 1u);
    }
}

void launch_bucket_finalize(hipStream_t s, const uint32_t* d_offs, uint32_t nb, uint32_t lanes, const void* d_part_a,
                            const void* d_part_b, void* d_buckets, void* d_heavy_ws, uint32_t* d_refs_out, uint32_t group) {
    HeavyWs ws = carve(d_heavy_ws);
    const uint4* pa = reinterpret_cast<const uint4*>(d_part_a);
    const uint4* pb = reinterpret_cast<const uint4*>(d_part_b);
    uint4* bk = reinterpret_cast<uint4*>(d_buckets);
    if (group >= 2) {
        const uint32_t per_block = 64 / group;
        hipLaunchKernelGGL(k_bucket_finalize_group, dim3((nb + per_block - 1) / per_block), dim3(64 * kCoop), 0, s, d_offs, nb,
                           lanes, pa, pb, bk, ws, d_refs_out, group);
    } else {
        hipLaunchKernelGGL(k_bucket_finalize, dim3((nb + 63) / 64), dim3(64 * kCoop), 0, s, d_offs, nb, lanes, pa, pb, bk, ws, d_refs_out);
    }
    hipLaunchKernelGGL(k_heavy_tree, dim3(kTreeGrid), dim3(kChunk * kCoop), 0, s, pa, pb, bk, ws);
}

uint32_t finalize_group_size(uint32_t nb) {
    // as many quads per bucket as ONE workgroup per CU offers (256 x 64 quads), 16 at most: a second workgroup on a CU
    // halves the speed of both, which costs more than the shorter chain of a larger group saves (2^17 terms, 16384
    // buckets of ~8 pieces: 85 us with two quads per bucket on 512 workgroups)
    uint32_t g = 1;
    while (g < 16 && (uint64_t)nb * g * 2 <= 16384) g <<= 1;
    return g;
}

uint32_t small_msm_lds_bytes() { return 84u * 1024u; }  // more than half a CU's 160 KB: one workgroup per CU
const void* small_msm_kernel() { return (const void*)k_small_msm; }

void launch_small_msm(hipStream_t s, const void* d_table, const uint32_t* d_sorted, const uint32_t* d_offs, uint32_t nb,
                      uint32_t lanes, uint64_t max_refs, void* d_buckets, void* d_part_a, void* d_part_b, void* d_heavy_ws,
                      uint32_t* d_refs_out, const TreeSumDesc* stage1, const TreeSumDesc* stage2, uint32_t lds_bytes) {
    SmallJob job;
    job.table = reinterpret_cast<const uint4*>(d_table);
    job.sorted = d_sorted;
    job.offs = d_offs;
    job.buckets = reinterpret_cast<uint4*>(d_buckets);
    job.part_a = reinterpret_cast<uint4*>(d_part_a);
    job.part_b = reinterpret_cast<uint4*>(d_part_b);
    job.refs_out = d_refs_out;
    job.ws = carve(d_heavy_ws);
    job.nb = nb;
    job.lanes = lanes;
    job.group = finalize_group_size(nb);
    // Very small jobs (an average bucket holds at most 32 references: degree <= 127 at the 8-bit width) skip the
    // accumulation phase: 2 serial additions + 4 tree levels per bucket instead of 3 + a phase boundary + 1 + 3.
    job.direct = max_refs <= (uint64_t)nb * 32 ? 1u : 0u;
    job.items_a = job.direct ? 0u : (lanes + 63) / 64;
    job.items_b = (nb + 64 / job.group - 1) / (64 / job.group);
    job.items_c = plan_tree_jobs(job.tj, stage1, 2, 0, 256u * kTreeLogical);
    job.items_d = plan_tree_jobs(job.tj + 2, stage2, 4, job.items_c, 256u * kTreeLogical);
    const uint32_t items = job.items_a + job.items_b + job.items_c + job.items_d;
    hipLaunchKernelGGL(k_small_msm, dim3(items < 256 ? items : 256), dim3(64 * kCoop), lds_bytes, s, job);
}

}  // namespace kzg
