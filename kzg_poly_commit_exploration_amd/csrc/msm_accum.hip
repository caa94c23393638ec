// msm_accum.hip -- bucket accumulation, the dominant kernel of the commitment.
//
// Replaces the reference's hot loop  commitment.add(srs[i].g1.mult(c_i))  (src/polynomial.rs:208-212):
// after msm_sort the (point, sign) references are sorted by bucket; their concatenation is cut into
// equal segments of L references and ONE LANE OWNS ONE SEGMENT, whatever buckets it crosses.  Every
// lane therefore performs the same number of mixed additions (XYZZ accumulator in VGPRs += affine
// table point, 8M + 2S of 384-bit Montgomery arithmetic on v_mad_u64_u32), and wavefronts stay
// converged for any scalar distribution -- uniform, the reference's i128-derived inputs, or all
// coefficients equal.  A lane closes a bucket it covers completely by storing it; the (at most two)
// buckets it shares with its neighbours leave a head / tail partial that k_bucket_finalize adds up.
//
// Roofline: VALU issue bound (a wave-level v_mad_u64_u32 costs ~5.1 issue cycles, a simple instruction ~3.0).  Per reference 2770 v_mad_u64_u32 + ~3800 other VALU ops
// against 96 B gathered from the table + 4 B of reference.  Algorithmic HBM bytes per commitment are
// those of SURVEY.md section 8(d): 128 B x n + 144 B.
#include <cstdlib>

#include "engine.h"
#include "g1.hip.h"

namespace kzg {

constexpr int kAccumBlock = 256;

KZG_DEV Affine load_affine(const uint4* __restrict__ table, uint32_t idx) {
    const uint4* p = table + (size_t)idx * kAffineU4;
    uint4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5];
    Affine r;
    r.x.l[0] = a0.x; r.x.l[1] = a0.y; r.x.l[2] = a0.z; r.x.l[3] = a0.w;
    r.x.l[4] = a1.x; r.x.l[5] = a1.y; r.x.l[6] = a1.z; r.x.l[7] = a1.w;
    r.x.l[8] = a2.x; r.x.l[9] = a2.y; r.x.l[10] = a2.z; r.x.l[11] = a2.w;
    r.y.l[0] = a3.x; r.y.l[1] = a3.y; r.y.l[2] = a3.z; r.y.l[3] = a3.w;
    r.y.l[4] = a4.x; r.y.l[5] = a4.y; r.y.l[6] = a4.z; r.y.l[7] = a4.w;
    r.y.l[8] = a5.x; r.y.l[9] = a5.y; r.y.l[10] = a5.z; r.y.l[11] = a5.w;
    return r;
}

KZG_DEV void store_xyzz(uint4* __restrict__ out, const XYZZ& a) {
    const Fp* f[4] = {&a.X, &a.Y, &a.ZZ, &a.ZZZ};
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int t = 0; t < 3; t++) {
            uint4 v;
            v.x = f[q]->l[4 * t];
            v.y = f[q]->l[4 * t + 1];
            v.z = f[q]->l[4 * t + 2];
            v.w = f[q]->l[4 * t + 3];
            out[q * 3 + t] = v;
        }
    }
}
[[maybe_unused]] KZG_DEV XYZZ load_xyzz(const uint4* __restrict__ in) {
    XYZZ a;
    Fp* f[4] = {&a.X, &a.Y, &a.ZZ, &a.ZZZ};
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int t = 0; t < 3; t++) {
            uint4 v = in[q * 3 + t];
            f[q]->l[4 * t] = v.x;
            f[q]->l[4 * t + 1] = v.y;
            f[q]->l[4 * t + 2] = v.z;
            f[q]->l[4 * t + 3] = v.w;
        }
    }
    return a;
}

// largest b in [0, nb) with offs[b] <= pos  (offs is non-decreasing, offs[0] = 0)
KZG_DEV uint32_t bucket_of(const uint32_t* __restrict__ offs, uint32_t nb, uint32_t pos) {
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
__global__ void __launch_bounds__(kAccumBlock) k_bucket_accumulate(const uint4* __restrict__ table,
                                                                  const uint32_t* __restrict__ sorted,
                                                                  const uint32_t* __restrict__ offs, uint32_t nb,
                                                                  uint32_t lanes, uint4* __restrict__ buckets,
                                                                  uint4* __restrict__ part_a,
                                                                  uint4* __restrict__ part_b) {
    const uint32_t lane = blockIdx.x * kAccumBlock + threadIdx.x;
    const uint32_t M = offs[nb];
    const uint32_t L = accumulate_seg_len(M, lanes);
    const uint64_t start64 = (uint64_t)lane * L;
    if (start64 >= M) return;
    const uint32_t start = (uint32_t)start64;
    const uint32_t end = (M - start < L) ? M : start + L;

    uint32_t b = bucket_of(offs, nb, start);
    uint32_t b_beg = offs[b], b_end = offs[b + 1];
    uint32_t run_start = start;
    XYZZ acc = XYZZ::inf();
    // software pipeline: the gather of reference e+1 is issued before the ~2900 multiply-adds of e
    // (+1.5 % measured).  Two points in flight, with the reference itself loaded an iteration earlier, measured
    // 2 % SLOWER (232 instead of 206 VGPRs, same 2 waves/SIMD): the gather latency is already covered.
    // Prefetching through LDS instead (global_load_lds, 188 VGPRs) measured the same as this, and forcing that
    // build to 168 VGPRs = 3 waves/SIMD did not make the kernel faster alone (2.94-2.99 vs 2.97-2.99 ms) and
    // starved the other slots' kernels (265 instead of 314 commitments/s): the kernel is bound by VALU issue
    // (VALUBusy 88 %), which two waves per SIMD already keep busy.
    u32 ref = sorted[start];
    Affine p = load_affine(table, ref & 0x7fffffffu);
    for (uint32_t e = start; e < end; e++) {
        if (e == b_end) {
            // bucket b ends here: flush its run and move to the bucket that owns e (skipping empties)
            uint4* dst = (run_start == b_beg) ? buckets + (size_t)b * 12 : part_a + (size_t)lane * 12;
            store_xyzz(dst, acc);  // a run that began inside the bucket necessarily began at `start`
            acc = XYZZ::inf();
            do {
                b++;
                b_beg = b_end;
                b_end = offs[b + 1];
            } while (b_end <= e);
            run_start = e;
        }
        const u32 cur_ref = ref;
        const Affine cur = p;
        if (e + 1 < end) {
            ref = sorted[e + 1];
            p = load_affine(table, ref & 0x7fffffffu);
        }
        xyzz_madd(acc, cur, (cur_ref >> 31) != 0);
    }
    // last run: [run_start, end)
    uint4* dst;
    if (run_start == b_beg && end == b_end) dst = buckets + (size_t)b * 12;  // complete
    else if (run_start == start) dst = part_a + (size_t)lane * 12;            // covers the whole segment
    else dst = part_b + (size_t)lane * 12;                                    // tail shared with the next lane
    store_xyzz(dst, acc);
}

uint32_t accumulate_lanes(uint64_t max_refs, bool alone) {
    // 196608 segments = 768 workgroups.  At 206 VGPRs two workgroups are resident per CU (512 at once); the
    // last 256 start as the first ones retire and then run with the SIMDs half empty, i.e. faster per wave.
    // Measured at 2^20 terms: 196608 lanes 3.16 ms, 131072 lanes (exactly one resident round) 3.21 ms,
    // 262144 lanes (two full rounds) are best ALONE (2.88 ms) and worst beside other slots' kernels (277 /s): see `alone`.
    // KZG_ACCUM_LANES overrides for experiments.
    static const uint64_t target = [] {
        const char* v = std::getenv("KZG_ACCUM_LANES");
        uint64_t l = v ? std::strtoull(v, nullptr, 10) : 196608ull;
        return l < 64 ? 262144ull : (l > 262144ull ? 262144ull : l);
    }();
    const uint64_t lo = accumulate_min_seg(max_refs);
    uint64_t lanes = (max_refs + lo - 1) / lo;  // segments are at least 8 (tiny jobs: 4) references long
    // alone on the chip (no other slot in flight) two full rounds are best: 2.88 instead of 2.96 ms at 2^20 terms
    const uint64_t cap = (alone && target == 196608ull) ? 262144ull : target;
    if (lanes > cap) lanes = cap;
    lanes = (lanes + kAccumBlock - 1) / kAccumBlock * kAccumBlock;
    return (uint32_t)lanes;
}

void launch_bucket_accumulate(hipStream_t s, const void* d_table, const uint32_t* d_sorted, const uint32_t* d_offs,
                              uint32_t nb, uint32_t lanes, void* d_buckets, void* d_part_a, void* d_part_b,
                              uint32_t lds_reserve_bytes) {
    if (!lanes) return;
    hipLaunchKernelGGL(k_bucket_accumulate, dim3(lanes / kAccumBlock), dim3(kAccumBlock), lds_reserve_bytes, s,
                       reinterpret_cast<const uint4*>(d_table), d_sorted, d_offs, nb, lanes,
                       reinterpret_cast<uint4*>(d_buckets), reinterpret_cast<uint4*>(d_part_a),
                       reinterpret_cast<uint4*>(d_part_b));
}

}  // namespace kzg
