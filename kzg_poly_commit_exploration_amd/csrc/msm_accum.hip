// msm_accum.hip -- bucket accumulation, the dominant kernel of the commitment.
//
// Replaces the reference's hot loop  commitment.add(srs[i].g1.mult(c_i))  (src/polynomial.rs:208-212):
// after msm_sort the (point, sign) references are sorted by bucket; their concatenation is cut into
// equal segments of L references and ONE LANE OWNS ONE SEGMENT, whatever buckets it crosses.  Every
// lane therefore performs the same number of mixed additions (XYZZ accumulator in VGPRs += affine
// table point, 8M + 2S in the signed radix-2^30 field of field30.hip.h: plain v_mad_i64_i32 chains), and wavefronts stay
// converged for any scalar distribution -- uniform, the reference's i128-derived inputs, or all
// coefficients equal.  A lane closes a bucket it covers completely by storing it; the (at most two)
// buckets it shares with its neighbours leave a head / tail partial that k_bucket_finalize adds up.
//
// Roofline: VALU issue bound.  Per mixed addition the wave executes 3055 v_mad_i64_i32 (6 products x 2 x 13^2, two squares
// x (91 + 13^2), and R (Q - X3) - Y1 PPP as two digit products under one reduction) + ~1400 other VALU instructions,
// against one 128-byte table record gathered by LDS-DMA + 4 B of reference; it runs within ~10 % of the sum of the two pipe
// times of that mix (measured bare-loop rates), at two waves per SIMD as well as at three -- a three-wave form of the kernel
// (no exceptional branches, 163 VGPRs, no spills) measured 2354 us against 2336 us alone, and a build of which three
// workgroups fit a CU is SLOWER in the pipeline (profiles/r03_accum_isa_histogram.txt).
// Algorithmic HBM bytes per commitment are those of SURVEY.md section 8(d): 128 B x n + 144 B.
#include <cstdlib>

// the products of this translation unit keep their columns in program order (one live accumulator, field30.hip.h): the same
// speed as the interleaved order at two waves per SIMD, and ~10 registers fewer for the kernels that run beside this one
#ifndef KZG_ACCUM_PARALLEL_COLUMNS
#define KZG_F30_SERIAL_COLUMNS 1
#endif
#include "engine.h"
#include "field30_inv.hip.h"
#include "g1_30.hip.h"

namespace kzg {

#ifndef KZG_ACCUM_BLOCK
#define KZG_ACCUM_BLOCK 256
#endif
constexpr int kAccumBlock = KZG_ACCUM_BLOCK;  // lanes per workgroup; the lanes never synchronise (the prefetch slots are per wave)

// table record (engine.h): x digits in words 0..12, y digits in words 16..28 of a 128-byte line
__device__ __forceinline__ Fq load_fq(const uint4* __restrict__ p) {
    const uint4 a = p[0], b = p[1], c = p[2];
    const uint32_t d = reinterpret_cast<const uint32_t*>(p)[12];
    Fq r;
    r.d[0] = (int32_t)a.x; r.d[1] = (int32_t)a.y; r.d[2] = (int32_t)a.z; r.d[3] = (int32_t)a.w;
    r.d[4] = (int32_t)b.x; r.d[5] = (int32_t)b.y; r.d[6] = (int32_t)b.z; r.d[7] = (int32_t)b.w;
    r.d[8] = (int32_t)c.x; r.d[9] = (int32_t)c.y; r.d[10] = (int32_t)c.z; r.d[11] = (int32_t)c.w;
    r.d[12] = (int32_t)d;
    return r;
}
__device__ __forceinline__ Affine30 load_affine30(const uint4* __restrict__ table, uint32_t idx) {
    const uint4* p = table + (size_t)idx * kAffineU4;
    Affine30 r;
    r.x = load_fq(p);
    r.y = load_fq(p + 4);
    return r;
}

// largest b in [0, nb) with offs[b] <= pos  (offs is non-decreasing, offs[0] = 0)
__device__ __forceinline__ uint32_t bucket_of(const uint32_t* __restrict__ offs, uint32_t nb, uint32_t pos) {
    uint32_t lo = 0, hi = nb;  // invariant: offs[lo] <= pos, (hi == nb or offs[hi] > pos)
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (offs[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

// One lane per segment [lane*L, lane*L + L) of the M sorted references, L = accumulate_seg_len(M, lanes):
// the launch only fixes the number of lanes, the segment length follows the actual number of non-zero digits.
//   complete runs   -> buckets[b]
//   run touching the segment start (bucket continues from the previous lane, or the whole segment
//   lies inside one bucket) -> part_a[lane];  run touching only the segment end -> part_b[lane]
// The prefetch slot of a wave: [piece 0..7][lane] uint4, 8 KiB of the launch's dynamic LDS.  Its LDS byte address is kept as a
// SCALAR (readfirstlane of a wave-uniform value the compiler cannot prove uniform): the LDS-DMA destination travels in M0, and
// from a vector register the compiler carries eight of them, one per piece, each read back into M0 -- and, once they are
// spilled, reloaded from scratch between the pieces behind a wait that serialises the gather.
typedef __attribute__((address_space(3))) char* lds_char_ptr;
typedef uint32_t accum_u32x4 __attribute__((ext_vector_type(4)));  // (uint4 has no address-space-qualified copy constructor)
typedef const __attribute__((address_space(3))) accum_u32x4* lds_u4_ptr;
extern __shared__ uint4 lds_points[];
// the lane's index in its wave from the execution hardware (two instructions, no register kept for it)
__device__ __forceinline__ uint32_t accum_wave_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t accum_wave_slot() {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_ptr)(char*)lds_points;
    return lds0 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * (8u * 64u * 16u);
}
// next point by LDS-DMA: eight 16-byte pieces per lane, destination = slot + piece * 1 KiB + lane * 16, no VGPR destination
__device__ __forceinline__ void accum_issue_gather(const uint4* __restrict__ table, uint32_t slot, uint32_t r) {
    const char* src = reinterpret_cast<const char*>(table + (size_t)(r & 0x7fffffffu) * kAffineU4);
#pragma unroll
    for (int k = 0; k < 8; k++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * k),
                                         (__attribute__((address_space(3))) void*)(lds_char_ptr)(uintptr_t)(slot + 1024u * k),
                                         16, 0, 0);
}
// kFlushOps: vector-memory operations a bucket flush issues between the gather and this wait (accum_flush_run's sixteen
// stores + the read of the next bucket's end).  The counter is in order for global_* loads, stores and LDS-DMA alike
// (MI355X_MICROARCH.md, "s_waitcnt vmcnt(N) waits until all but the wave's N youngest vector-memory operations are done"; only
// flat_* is excepted, and there is none here): "at most kFlushOps outstanding" means the gather and the reference load in
// front of them have landed, without waiting for a store's round trip to come back (+2 % at degree
// 2^20, where one iteration in four of a wave has a lane at a bucket's end).  A SMALLER number of operations on that path
// would make the wait too weak (the point would be read before it has arrived): tests/test_accum_isa.py counts them in
// the compiler's output.  (The walk over empty buckets waits for everything inside its loop, so behind it the number is
// smaller and the gather has landed anyway.)
constexpr int kFlushOps = 17;
static_assert(kFlushOps == 17, "the literal in accum_take_point's s_waitcnt");
__device__ __forceinline__ Affine30 accum_take_point(uint32_t slot, bool wave_flushed = false) {
#ifdef KZG_ACCUM_FULL_WAIT  // (A/B: wait for the flush too)
    wave_flushed = false;
#endif
    if (wave_flushed) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA pieces (and the reference load behind them)
    Affine30 r;
    lds_u4_ptr q = (lds_u4_ptr)(uintptr_t)(slot + accum_wave_lane() * 16u);
    const accum_u32x4 x0 = q[0], x1 = q[64], x2 = q[128], x3 = q[192];
    const accum_u32x4 y0 = q[256], y1 = q[320], y2 = q[384], y3 = q[448];
    r.x.d[0] = (int32_t)x0.x; r.x.d[1] = (int32_t)x0.y; r.x.d[2] = (int32_t)x0.z; r.x.d[3] = (int32_t)x0.w;
    r.x.d[4] = (int32_t)x1.x; r.x.d[5] = (int32_t)x1.y; r.x.d[6] = (int32_t)x1.z; r.x.d[7] = (int32_t)x1.w;
    r.x.d[8] = (int32_t)x2.x; r.x.d[9] = (int32_t)x2.y; r.x.d[10] = (int32_t)x2.z; r.x.d[11] = (int32_t)x2.w;
    r.x.d[12] = (int32_t)x3.x;
    r.y.d[0] = (int32_t)y0.x; r.y.d[1] = (int32_t)y0.y; r.y.d[2] = (int32_t)y0.z; r.y.d[3] = (int32_t)y0.w;
    r.y.d[4] = (int32_t)y1.x; r.y.d[5] = (int32_t)y1.y; r.y.d[6] = (int32_t)y1.z; r.y.d[7] = (int32_t)y1.w;
    r.y.d[8] = (int32_t)y2.x; r.y.d[9] = (int32_t)y2.y; r.y.d[10] = (int32_t)y2.z; r.y.d[11] = (int32_t)y2.w;
    r.y.d[12] = (int32_t)y3.x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot may be overwritten from here on
    return r;
}
// A finished run leaves the lane: exactly SIXTEEN 16-byte stores, issued as one block the compiler neither splits, merges nor
// counts (store_xyzz30 compiles to 17 of mixed widths today; accum_take_point's wait depends on the number).
__device__ __forceinline__ void accum_flush_run(uint4* dst, const XYZZ30& a) {
    const Fq* f[4] = {&a.X, &a.Y, &a.ZZ, &a.ZZZ};
    accum_u32x4 v[16];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        v[4 * c + 0] = accum_u32x4{(uint32_t)f[c]->d[0], (uint32_t)f[c]->d[1], (uint32_t)f[c]->d[2], (uint32_t)f[c]->d[3]};
        v[4 * c + 1] = accum_u32x4{(uint32_t)f[c]->d[4], (uint32_t)f[c]->d[5], (uint32_t)f[c]->d[6], (uint32_t)f[c]->d[7]};
        v[4 * c + 2] = accum_u32x4{(uint32_t)f[c]->d[8], (uint32_t)f[c]->d[9], (uint32_t)f[c]->d[10], (uint32_t)f[c]->d[11]};
        v[4 * c + 3] = accum_u32x4{(uint32_t)f[c]->d[12], 0u, 0u, 0u};
    }
    asm volatile(
        "global_store_dwordx4 %0, %1, off\n\t"
        "global_store_dwordx4 %0, %2, off offset:16\n\t"
        "global_store_dwordx4 %0, %3, off offset:32\n\t"
        "global_store_dwordx4 %0, %4, off offset:48\n\t"
        "global_store_dwordx4 %0, %5, off offset:64\n\t"
        "global_store_dwordx4 %0, %6, off offset:80\n\t"
        "global_store_dwordx4 %0, %7, off offset:96\n\t"
        "global_store_dwordx4 %0, %8, off offset:112\n\t"
        "global_store_dwordx4 %0, %9, off offset:128\n\t"
        "global_store_dwordx4 %0, %10, off offset:144\n\t"
        "global_store_dwordx4 %0, %11, off offset:160\n\t"
        "global_store_dwordx4 %0, %12, off offset:176\n\t"
        "global_store_dwordx4 %0, %13, off offset:192\n\t"
        "global_store_dwordx4 %0, %14, off offset:208\n\t"
        "global_store_dwordx4 %0, %15, off offset:224\n\t"
        "global_store_dwordx4 %0, %16, off offset:240"
        :
        : "v"(dst), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
          "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15])
        : "memory");
}
// sorted[i] with a 32-bit byte offset on the scalar base (the references of one job are < 2^30)
__device__ __forceinline__ uint32_t accum_ref(const uint32_t* __restrict__ sorted, uint32_t i) {
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(sorted) + (uint32_t)(i * 4u));
}

#ifndef KZG_ACCUM_MIN_BLOCKS
#define KZG_ACCUM_MIN_BLOCKS 2
#endif
__global__ void __launch_bounds__(kAccumBlock, KZG_ACCUM_MIN_BLOCKS)
k_bucket_accumulate(const uint4* __restrict__ table, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offs,
                    uint32_t nb, uint32_t lanes, uint4* __restrict__ buckets, uint4* __restrict__ part_a,
                    uint4* __restrict__ part_b, unsigned long long* __restrict__ clk) {
    // clk (may be null): two zeroed words that receive max(~start) and max(end) over the waves of the launch, in ticks
    // of the constant 100 MHz clock -- the kernel's own duration, measured without a stream event on either side of it
    // (events between consecutive accumulation kernels cost the pipeline 8-10 %)
    // At most TWO workgroups of this kernel per CU, by register count (the clobber reserves v0..v175: 2 x 176 of the 512
    // registers of a SIMD lane fit, 3 do not).  The grid is exactly two workgroups per CU; a build that fits three (<= 168
    // VGPRs) lets the dispatcher stack three on the CUs that are free when the launch begins -- beside the light kernels of
    // the other slots some always are busy -- and the launch then lasts as long as its most crowded CU: 2.96 ms instead of
    // 2.32 in the pipeline, with the same 2.29 ms alone.
    asm volatile("" ::: "v175");
    const bool stamp = clk != nullptr && (threadIdx.x & 63) == 0;
    if (stamp) atomicMax(&clk[0], ~(unsigned long long)wall_clock64());
    const uint32_t lane = blockIdx.x * kAccumBlock + threadIdx.x;
    const uint32_t M = offs[nb];
    const uint32_t L = accumulate_seg_len(M, lanes);
    const uint64_t start64 = (uint64_t)lane * L;
    if (start64 >= M) return;
    const uint32_t start = (uint32_t)start64;
    const uint32_t end = (M - start < L) ? M : start + L;

    uint32_t b = bucket_of(offs, nb, start);
    uint32_t b_beg = offs[b], b_end = offs[b + 1];
    uint32_t b_end_next = offs[(b + 2 <= nb) ? b + 2 : nb];
    uint32_t run_start = start;
    XYZZ30 acc = xyzz30_inf();
    // Software pipeline through LDS.  The point of reference e is consumed by the first two products of its addition
    // (P, R); the gather of reference e+1 is issued right behind them as eight 16-byte LDS-DMA pieces
    // (global_load_lds_dwordx4: per-lane source address, destination = wave base + lane * 16, no VGPR destination), so
    // the 26 registers of a point are live only across those two products instead of the whole addition.  The DMA has
    // the remaining 6M + 2S (~3500 instructions) to land; the table reference itself is read one step further ahead.
    const uint32_t wave_slot = accum_wave_slot();  // [piece 0..7][lane]: 8 KiB per wave (dynamic LDS of the launch)
    uint32_t ref = accum_ref(sorted, start);
    uint32_t ref_next = accum_ref(sorted, min(start + 1, M - 1));
    accum_issue_gather(table, wave_slot, ref);
    for (uint32_t e = start; e < end; e++) {
        // (wave-uniform: the flush below is issued for the wave when ANY lane is at a bucket's end)
        const bool wave_flushed = __builtin_amdgcn_ballot_w64(e == b_end) != 0;
        if (e == b_end) {
            // bucket b ends here: flush its run and move to the bucket that owns e (skipping empties)
            uint4* dst = (run_start == b_beg) ? buckets + (size_t)b * kXyzzU4 : part_a + (size_t)lane * kXyzzU4;
            accum_flush_run(dst, acc);  // a run that began inside the bucket necessarily began at `start`
            acc = xyzz30_inf();
            // the end of the next bucket was read one boundary ago: the walk does not wait for memory unless it has
            // to skip empty buckets (then its own waits take the stores above with them: rare)
            b++;
            b_beg = b_end;
            b_end = b_end_next;
            while (b_end <= e) {
                b++;
                b_beg = b_end;
                b_end = offs[b + 1];
            }
            b_end_next = offs[(b + 2 <= nb) ? b + 2 : nb];
            run_start = e;
        }
        Fq P, R;
        bool more;
        {
            // After a flush the wait lets the flush's own kFlushOps operations stay in flight: without that every bucket
            // boundary in a wave -- one iteration in four at degree 2^20, nine in ten in a batch of short polynomials --
            // stood still for a store's round trip before it looked at its point.
            const Affine30 p = accum_take_point(wave_slot, wave_flushed);
            more = xyzz30_madd_head(acc, p, (ref >> 31) != 0, P, R);
        }
        // The next reference and the load of the one after it are UNCONDITIONAL (the index clamped to the job's last
        // reference; past the segment's end the values are not used): assigned under a condition, the loaded word is
        // merged with the old one by a move right here -- behind a wait that exposes the gather's whole latency.
        ref = ref_next;
        if (e + 1 < end) accum_issue_gather(table, wave_slot, ref);
        ref_next = accum_ref(sorted, min(e + 2, M - 1));  // waited for at the next point, an addition from here
        if (more) xyzz30_madd_tail(acc, P, R);
    }
    // last run: [run_start, end)
    uint4* dst;
    if (run_start == b_beg && end == b_end) dst = buckets + (size_t)b * kXyzzU4;  // complete
    else if (run_start == start) dst = part_a + (size_t)lane * kXyzzU4;            // covers the whole segment
    else dst = part_b + (size_t)lane * kXyzzU4;                                    // tail shared with the next lane
    store_xyzz30(dst, acc);
    if (stamp) atomicMax(&clk[1], (unsigned long long)wall_clock64());
}

// ---- the same with an affine front end -----------------------------------------------------------------------------
// Two consecutive references of a segment that lie in the same bucket are first added in affine coordinates (2M + 1S
// once the inverse of x2 - x1 is known) and their sum enters the XYZZ accumulator with ONE mixed addition instead of
// two: per pair 1 (prefix product) + 2 (Montgomery's trick) + 2M + 1S + one mixed addition ~ 15.5 field products
// instead of 19.4, for one shared inversion per lane (safegcd, ~40 products, field30_inv.hip.h) and a second gather of
// the two points.
//   forward pass   pair slot j = references (start + 2j, start + 2j + 1): classify (pair_classify), store the running
//                  product of the denominators so far and the kind (64-byte record, lane-interleaved: coalesced);
//   inversion      of the product of all denominators of the lane;
//   backward pass  slots in descending order: 1/den_j = inv * prefix_j, inv *= den_j; affine sum; mixed addition into
//                  the accumulator with the run logic of k_bucket_accumulate mirrored (runs flushed when the walk
//                  crosses the lower end of a bucket).  Slots that are not pairs are walked as single references.
// Exceptional pairs (equal points, opposite points, infinity) are classified in the forward pass and never poison the
// shared product: their denominator is 1 (or 2y for a doubling).  Outputs are identical to k_bucket_accumulate's.
constexpr uint32_t kPairRecU4 = 4;  // prefix record: 13 digits + kind, padded to 64 bytes

// field products of the pair arithmetic as real calls (arguments and result in registers): five inlined products
// more would push the backward loop past the instruction cache (73 KB measured, 40 % slower than without pairs)
static __device__ __noinline__ Fq fq_mul_call(Fq a, Fq b) { return fq_mul(a, b); }
static __device__ __noinline__ Fq fq_sqr_call(Fq a) { return fq_sqr(a); }
#ifdef KZG_PAIR_CALLS  // A/B switch: the five products of the pair arithmetic as calls or inlined
#define KZG_PAIR_MUL(a, b) fq_mul_call(a, b)
#define KZG_PAIR_SQR(a) fq_sqr_call(a)
#else
#define KZG_PAIR_MUL(a, b) fq_mul(a, b)
#define KZG_PAIR_SQR(a) fq_sqr(a)
#endif
// the inversion runs once per lane: a real call keeps its ~100 registers of state out of the loops' allocation
static __device__ __noinline__ void fq_inv_call(Fq* io) {
    const Fq a = *io;
    *io = fq_inv(a);
}
// rare path of the forward pass (equal / opposite / infinite points): a real call, operands through private memory
static __device__ __noinline__ uint32_t pair_classify_call(const Affine30* a, bool nega, const Affine30* b, bool negb, Fq* den) {
    Fq d;
    const uint32_t kind = pair_classify(*a, nega, *b, negb, d);
    *den = d;
    return kind;
}

__global__ void __launch_bounds__(kAccumBlock, 2) k_bucket_accumulate_pairs(const uint4* __restrict__ table,
                                                                            const uint32_t* __restrict__ sorted,
                                                                            const uint32_t* __restrict__ offs, uint32_t nb,
                                                                            uint32_t lanes, uint4* __restrict__ buckets,
                                                                            uint4* __restrict__ part_a,
                                                                            uint4* __restrict__ part_b,
                                                                            uint4* __restrict__ prefix_buf) {
    const uint32_t lane = blockIdx.x * kAccumBlock + threadIdx.x;
    const uint32_t M = offs[nb];
    const uint32_t L = accumulate_seg_len(M, lanes);
    const uint64_t start64 = (uint64_t)lane * L;
    if (start64 >= M) return;
    const uint32_t start = (uint32_t)start64;
    const uint32_t end = (M - start < L) ? M : start + L;
    const uint32_t J = (end - start + 1) / 2;  // pair slots of this lane (the last one may hold a single reference)
    // record (j, piece) of this lane: ((j * 4 + piece) * lanes + lane): a wave touches 1 KiB contiguous per piece
    auto rec = [&](uint32_t j, uint32_t piece) { return prefix_buf + ((size_t)(j * kPairRecU4 + piece) * lanes + lane); };
    // LDS: two point slots per wave (first and second reference of a pair slot), 8 pieces x 64 lanes x 16 B each
    uint4* const slot0 = lds_points + (threadIdx.x >> 6) * (16 * 64);
    uint4* const slot1 = slot0 + 8 * 64;
    const uint32_t wl = threadIdx.x & 63;
    auto dma = [&](uint4* slot, uint32_t r, int first_piece, int pieces) {
        const char* src = reinterpret_cast<const char*>(table + (size_t)(r & 0x7fffffffu) * kAffineU4);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k >= first_piece && k < first_piece + pieces)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * k),
                                                 (__attribute__((address_space(3))) void*)(slot + 64 * k), 16, 0, 0);
    };
    auto lds_fq = [&](const uint4* slot, int first_piece) {
        const uint4 a = slot[64 * first_piece + wl], b2 = slot[64 * (first_piece + 1) + wl], c = slot[64 * (first_piece + 2) + wl],
                    d = slot[64 * (first_piece + 3) + wl];
        Fq r;
        r.d[0] = (int32_t)a.x; r.d[1] = (int32_t)a.y; r.d[2] = (int32_t)a.z; r.d[3] = (int32_t)a.w;
        r.d[4] = (int32_t)b2.x; r.d[5] = (int32_t)b2.y; r.d[6] = (int32_t)b2.z; r.d[7] = (int32_t)b2.w;
        r.d[8] = (int32_t)c.x; r.d[9] = (int32_t)c.y; r.d[10] = (int32_t)c.z; r.d[11] = (int32_t)c.w;
        r.d[12] = (int32_t)d.x;
        return r;
    };

    // ---------------- forward pass: slot j's x coordinates arrive by LDS-DMA while slot j-1 is multiplied
    uint32_t b = bucket_of(offs, nb, start);
    uint32_t b_end = offs[b + 1];
    Fq run = fq_one();
    {
        uint32_t r0 = sorted[start], r1 = start + 1 < end ? sorted[start + 1] : 0u;
        uint32_t r0n = start + 2 < end ? sorted[start + 2] : 0u, r1n = start + 3 < end ? sorted[start + 3] : 0u;
        dma(slot0, r0, 0, 4);
        dma(slot1, r1, 0, 4);
        for (uint32_t j = 0; j < J; j++) {
            const uint32_t e0 = start + 2 * j, e1 = e0 + 1;
            while (b_end <= e0) {
                b++;
                b_end = offs[b + 1];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const Fq x0 = lds_fq(slot0, 0), x1 = lds_fq(slot1, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t c0 = r0, c1 = r1;
            if (j + 1 < J) {  // next slot's x coordinates, and the references of the slot after it
                r0 = r0n;
                r1 = r1n;
                dma(slot0, r0, 0, 4);
                dma(slot1, r1, 0, 4);
                r0n = e0 + 4 < end ? sorted[e0 + 4] : 0u;
                r1n = e0 + 5 < end ? sorted[e0 + 5] : 0u;
            }
            uint32_t kind = kPairNone;
            Fq den = fq_one();
            if (e1 < end && e1 < b_end) {
                const Fq dx = fq_norm(fq_sub_raw(x1, x0));
                if (!fq_all_zero(x0) && !fq_all_zero(x1) && !fq_is_zero(dx)) {
                    kind = kPairAdd;  // the common case needs the x coordinates only
                    den = dx;
                } else {
                    Affine30 a, c;
                    a.x = x0;
                    a.y = load_fq(table + (size_t)(c0 & 0x7fffffffu) * kAffineU4 + 4);
                    c.x = x1;
                    c.y = load_fq(table + (size_t)(c1 & 0x7fffffffu) * kAffineU4 + 4);
                    kind = pair_classify_call(&a, (c0 >> 31) != 0, &c, (c1 >> 31) != 0, &den);
                }
            }
            *rec(j, 0) = make_uint4((uint32_t)run.d[0], (uint32_t)run.d[1], (uint32_t)run.d[2], (uint32_t)run.d[3]);
            *rec(j, 1) = make_uint4((uint32_t)run.d[4], (uint32_t)run.d[5], (uint32_t)run.d[6], (uint32_t)run.d[7]);
            *rec(j, 2) = make_uint4((uint32_t)run.d[8], (uint32_t)run.d[9], (uint32_t)run.d[10], (uint32_t)run.d[11]);
            *rec(j, 3) = make_uint4((uint32_t)run.d[12], kind, c0, c1);
            if (kind == kPairAdd || kind == kPairDouble) run = KZG_PAIR_MUL(run, den);
        }
    }
    // ---------------- one inversion for the lane
    Fq inv = run;
    fq_inv_call(&inv);

    // ---------------- backward pass: ONE walk over the references, downwards, one mixed addition per step for
    // every lane (a pair slot is consumed when the walk reaches its second reference; a slot without a pair is
    // walked as single references).  Lanes keep their own cursor, so a lane that meets a bucket boundary inside a
    // slot does not cost its wave an extra addition.  Prefetch: the record of the next step's slot (plain loads) and
    // the points at the next cursor and below it (LDS-DMA) are requested as soon as this step's kind is known.
    b = bucket_of(offs, nb, end - 1);
    uint32_t b_beg = offs[b];
    b_end = offs[b + 1];
    XYZZ30 acc = xyzz30_inf();
    auto flush = [&]() {
        // the run of bucket b inside [start, end)
        uint4* dst;
        if (b_beg >= start && b_end <= end) dst = buckets + (size_t)b * kXyzzU4;  // complete
        else if (b_beg <= start) dst = part_a + (size_t)lane * kXyzzU4;           // touches the segment start
        else dst = part_b + (size_t)lane * kXyzzU4;                                // touches only its end
        store_xyzz30(dst, acc);
    };
    auto enter = [&](uint32_t e) {  // make b the bucket of reference e (walking downwards)
        if (e < b_beg) {
            flush();
            acc = xyzz30_inf();
            do {
                b--;
                b_end = b_beg;
                b_beg = offs[b];
            } while (b_beg > e);
        }
    };
    uint32_t e = end - 1;                      // cursor: the reference this step starts from
    // upcoming references below the cursor: w0 = sorted[e], w1 = sorted[e-1], w2 = sorted[e-2], w3 = sorted[e-3]
    auto ref_at = [&](uint32_t idx_plus_1) { return idx_plus_1 > start ? sorted[idx_plus_1 - 1] : 0u; };
    uint32_t w0 = sorted[e], w1 = ref_at(e), w2 = e >= 1 ? ref_at(e - 1) : 0u, w3 = e >= 2 ? ref_at(e - 2) : 0u;
    uint4 q0, q1, q2, q3;
    {
        const uint32_t j = (e - start) >> 1;
        q0 = *rec(j, 0); q1 = *rec(j, 1); q2 = *rec(j, 2); q3 = *rec(j, 3);
        dma(slot1, w0, 0, 8);  // point at the cursor
        dma(slot0, w1, 0, 8);  // point below it (used when the step is a pair)
    }
    bool more_steps = true;
    while (more_steps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t j = (e - start) >> 1;
        const uint32_t kind = q3.y;
        const bool second = ((e - start) & 1u) != 0;
        const bool paired = kind != kPairNone && second;  // (a slot of another kind is always entered at its second reference)
        const uint32_t r_hi = w0, r_lo = w1;              // references at e and e - 1
        Fq prefix;
        prefix.d[0] = (int32_t)q0.x; prefix.d[1] = (int32_t)q0.y; prefix.d[2] = (int32_t)q0.z; prefix.d[3] = (int32_t)q0.w;
        prefix.d[4] = (int32_t)q1.x; prefix.d[5] = (int32_t)q1.y; prefix.d[6] = (int32_t)q1.z; prefix.d[7] = (int32_t)q1.w;
        prefix.d[8] = (int32_t)q2.x; prefix.d[9] = (int32_t)q2.y; prefix.d[10] = (int32_t)q2.z; prefix.d[11] = (int32_t)q2.w;
        prefix.d[12] = (int32_t)q3.x;
        Affine30 pt;  // the point at the cursor
        pt.x = lds_fq(slot1, 0);
        pt.y = lds_fq(slot1, 4);
        Affine30 lo;  // the point below it
        lo.x = lds_fq(slot0, 0);
        lo.y = lds_fq(slot0, 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        enter(e);
        // next step's cursor, its record and its points
        const uint32_t used = paired ? 2u : 1u;
        more_steps = e >= start + used;
        if (more_steps) {
            e -= used;
            if (paired) { w0 = w2; w1 = w3; } else { w0 = w1; w1 = w2; w2 = w3; }
            const uint32_t jn = (e - start) >> 1;
            q0 = *rec(jn, 0); q1 = *rec(jn, 1); q2 = *rec(jn, 2); q3 = *rec(jn, 3);
            dma(slot1, w0, 0, 8);
            dma(slot0, w1, 0, 8);
            // refill the window: references at e - 2 and e - 3 (those at e and e - 1 are in w0, w1)
            if (paired) {
                w2 = e >= start + 2 ? sorted[e - 2] : 0u;
                w3 = e >= start + 3 ? sorted[e - 3] : 0u;
            } else {
                w3 = e >= start + 3 ? sorted[e - 3] : 0u;
            }
        }
        bool neg = (r_hi >> 31) != 0;
        bool have = true;
        if (paired) {
            const bool n0 = (r_lo >> 31) != 0;
            if (kind == kPairCancel) {
                have = false;
            } else if (kind == kPairOnlyA) {
                pt = lo;
                neg = n0;
            } else if (kind != kPairOnlyB) {
                const Fq den = kind == kPairAdd ? fq_norm(fq_sub_raw(pt.x, lo.x)) : fq_norm(fq_add_raw(fq_cneg(lo.y, n0), fq_cneg(pt.y, neg)));
                const Fq inv_den = KZG_PAIR_MUL(inv, prefix);
                KZG_SB30();
                inv = KZG_PAIR_MUL(inv, den);
                KZG_SB30();
                pt = pair_sum_with(kind, lo, n0, pt, neg, inv_den, [](const Fq& x, const Fq& y) { return KZG_PAIR_MUL(x, y); },
                                   [](const Fq& x) { return KZG_PAIR_SQR(x); });
                neg = false;
            }
        }
        if (have) xyzz30_madd(acc, pt, neg);
    }
    flush();
}

// ---- table format -------------------------------------------------------------------------------------------------
// native records -> blst_p1 (canonical 12 x u32 Montgomery R = 2^384 limbs, Z = one or zero): kzg_srs_read_g1
__global__ void __launch_bounds__(64) k_fq_table_to_p1(const uint4* __restrict__ table, uint32_t n, uint4* __restrict__ out) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const Affine30 a = load_affine30(table, i);
    uint32_t w[36];
#pragma unroll
    for (int t = 0; t < 36; t++) w[t] = 0;
    if (!(fq_all_zero(a.x) && fq_all_zero(a.y))) {
        fq_to_u32x12(a.x, w);
        fq_to_u32x12(a.y, w + 12);
        constexpr uint32_t ONE[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                      0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
#pragma unroll
        for (int t = 0; t < 12; t++) w[24 + t] = ONE[t];
    }
    uint4* o = out + (size_t)i * 9;
#pragma unroll
    for (int t = 0; t < 9; t++) o[t] = make_uint4(w[4 * t], w[4 * t + 1], w[4 * t + 2], w[4 * t + 3]);
}
void launch_affine_to_p1(hipStream_t s, const void* d_affine, uint32_t n, void* d_p1) {
    if (!n) return;
    hipLaunchKernelGGL(k_fq_table_to_p1, dim3((n + 63) / 64), dim3(64), 0, s, (const uint4*)d_affine, n, (uint4*)d_p1);
}

uint32_t accumulate_lanes(uint64_t max_refs, bool alone) {
    // 131072 segments = 512 workgroups = exactly the one round that is resident at 256 VGPRs (two workgroups per
    // CU).  Measured at 2^20 terms with three slots in flight (round 2, signed radix-2^30 field), two boxes:
    // 131072 lanes 2.64 / 2.79 ms per launch and 340 / 333 commitments/s, 196608 lanes 2.76 / 2.82 ms and 327 / 327,
    // 262144 lanes 2.60 / 2.98 ms and 342 / 321: one resident round is the robust choice.  (Round 1's 206-VGPR kernel
    // left room for the light kernels of other slots beside it and preferred 196608 lanes; this one fills the
    // register file.)  KZG_ACCUM_LANES overrides for experiments.
    (void)alone;
    static const uint64_t target = [] {
        const char* v = std::getenv("KZG_ACCUM_LANES");
        uint64_t l = v ? std::strtoull(v, nullptr, 10) : 131072ull;
        return l < 64 ? 131072ull : (l > 262144ull ? 262144ull : l);
    }();
    const uint64_t lo = accumulate_min_seg(max_refs);
    uint64_t lanes = (max_refs + lo - 1) / lo;  // segments are at least 8 (tiny jobs: 4) references long
    if (lanes > target) lanes = target;
    lanes = (lanes + kAccumBlock - 1) / kAccumBlock * kAccumBlock;
    return (uint32_t)lanes;
}

// KZG_ACCUM_PAIRS=1 selects the affine front end (segments of at least 16 references).  Off by default: bit-exact
// (the GPU suite runs a child process under it) but SLOWER on MI355X -- measured at 2^20 terms, same box, 131072 lanes:
// 3.66 ms per launch against 2.81 ms for plain mixed additions (3.20 ms without any prefetch, 3.85 ms with a
// per-slot loop that let one lane's bucket boundary cost its wave an extra addition, 4.0-4.2 ms with the five pair
// products as real calls).  The 3.9 field products a pair saves are eaten by what the product count does not show:
// two carry passes and a zero test per affine sum, the prefix records, a second gather, the 64 + 16 LDS-DMA /
// LDS reads per step, a loop body of 73 KB against a 64 KB instruction cache, and ~50 products' worth of
// strictly serial division steps per lane for the shared inversion.
bool accumulate_pairs_enabled() {
    static const int pairs_mode = [] {
        const char* v = std::getenv("KZG_ACCUM_PAIRS");
        return v ? std::atoi(v) : 0;
    }();
    return pairs_mode != 0;
}
size_t accumulate_pair_scratch_bytes(uint64_t max_refs) {
    // one 64-byte record per pair slot: at most max_refs / 2 + one per lane; nothing when the front end is off
    if (!accumulate_pairs_enabled()) return 0;
    return (size_t)(max_refs / 2 + kMaxAccumLanes) * kPairRecU4 * 16;
}

void launch_bucket_accumulate(hipStream_t s, const void* d_table, const uint32_t* d_sorted, const uint32_t* d_offs,
                              uint32_t nb, uint32_t lanes, void* d_buckets, void* d_part_a, void* d_part_b,
                              uint32_t lds_reserve_bytes, void* d_pair_scratch, uint64_t max_refs, void* d_clock) {
    if (!lanes) return;
    if (accumulate_pairs_enabled() && d_pair_scratch && max_refs / lanes >= 16) {
        hipLaunchKernelGGL(k_bucket_accumulate_pairs, dim3(lanes / kAccumBlock), dim3(kAccumBlock),
                           (kAccumBlock / 64) * 16 * 64 * 16 /* two point slots per wave */, s,
                           reinterpret_cast<const uint4*>(d_table), d_sorted, d_offs, nb, lanes,
                           reinterpret_cast<uint4*>(d_buckets), reinterpret_cast<uint4*>(d_part_a),
                           reinterpret_cast<uint4*>(d_part_b), reinterpret_cast<uint4*>(d_pair_scratch));
        return;
    }
    // dynamic LDS: the prefetch slots (8 KiB per wave), or more when the caller reserves LDS to shape occupancy
    const uint32_t need = (kAccumBlock / 64) * 8 * 64 * 16;
    lds_reserve_bytes = lds_reserve_bytes / (256 / kAccumBlock);  // the caller's reservation is per 256 lanes
    if (lds_reserve_bytes < need) lds_reserve_bytes = need;
    hipLaunchKernelGGL(k_bucket_accumulate, dim3(lanes / kAccumBlock), dim3(kAccumBlock), lds_reserve_bytes, s,
                       reinterpret_cast<const uint4*>(d_table), d_sorted, d_offs, nb, lanes,
                       reinterpret_cast<uint4*>(d_buckets), reinterpret_cast<uint4*>(d_part_a),
                       reinterpret_cast<uint4*>(d_part_b), reinterpret_cast<unsigned long long*>(d_clock));
}

}  // namespace kzg
