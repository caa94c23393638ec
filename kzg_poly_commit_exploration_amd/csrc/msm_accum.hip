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
// Roofline: VALU issue bound (a wave-level v_mad_u64_u32 costs ~5.1 issue cycles, a simple instruction ~3.0).  Per reference 2770 v_mad_u64_u32 + ~3800 other VALU ops
// against 96 B gathered from the table + 4 B of reference.  Algorithmic HBM bytes per commitment are
// those of SURVEY.md section 8(d): 128 B x n + 144 B.
#include <cstdlib>

#include "engine.h"
#include "g1_30.hip.h"

namespace kzg {

constexpr int kAccumBlock = 256;

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
#ifdef KZG_ACCUM_VGPRS  // experiments: cap the allocation below the 256 that two waves per SIMD allow
#define KZG_ACCUM_ATTR __attribute__((amdgpu_num_vgpr(KZG_ACCUM_VGPRS)))
#else
#define KZG_ACCUM_ATTR
#endif
__global__ void KZG_ACCUM_ATTR __launch_bounds__(kAccumBlock, 2) k_bucket_accumulate(const uint4* __restrict__ table,
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
    XYZZ30 acc = xyzz30_inf();
    // Software pipeline through LDS.  The point of reference e is consumed by the first two products of its addition
    // (P, R); the gather of reference e+1 is issued right behind them as eight 16-byte LDS-DMA pieces
    // (global_load_lds_dwordx4: per-lane source address, destination = wave base + lane * 16, no VGPR destination), so
    // the 26 registers of a point are live only across those two products instead of the whole addition.  The DMA has
    // the remaining 6M + 2S (~3500 instructions) to land; the table reference itself is read one step further ahead.
    extern __shared__ uint4 lds_points[];  // [wave][piece 0..7][lane]: 8 KiB per wave (dynamic LDS of the launch)
    uint4* const wave_slot = lds_points + (threadIdx.x >> 6) * (8 * 64);
    const uint32_t wl = threadIdx.x & 63;
    auto issue_gather = [&](uint32_t r) {
        const char* src = reinterpret_cast<const char*>(table + (size_t)(r & 0x7fffffffu) * kAffineU4);
#pragma unroll
        for (int k = 0; k < 8; k++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * k),
                                             (__attribute__((address_space(3))) void*)(wave_slot + 64 * k), 16, 0, 0);
    };
    auto take_point = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA pieces (and the reference load behind them)
        Affine30 r;
        const uint4 x0 = wave_slot[wl], x1 = wave_slot[64 + wl], x2 = wave_slot[128 + wl], x3 = wave_slot[192 + wl];
        const uint4 y0 = wave_slot[256 + wl], y1 = wave_slot[320 + wl], y2 = wave_slot[384 + wl], y3 = wave_slot[448 + wl];
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
    };
    uint32_t ref = sorted[start];
    uint32_t ref_next = start + 1 < end ? sorted[start + 1] : 0u;
    issue_gather(ref);
    for (uint32_t e = start; e < end; e++) {
        if (e == b_end) {
            // bucket b ends here: flush its run and move to the bucket that owns e (skipping empties)
            uint4* dst = (run_start == b_beg) ? buckets + (size_t)b * kXyzzU4 : part_a + (size_t)lane * kXyzzU4;
            store_xyzz30(dst, acc);  // a run that began inside the bucket necessarily began at `start`
            acc = xyzz30_inf();
            do {
                b++;
                b_beg = b_end;
                b_end = offs[b + 1];
            } while (b_end <= e);
            run_start = e;
        }
        Fq P, R;
        bool more;
        {
            const Affine30 p = take_point();
            more = xyzz30_madd_head(acc, p, (ref >> 31) != 0, P, R);
        }
        if (e + 1 < end) {
            ref = ref_next;
            issue_gather(ref);
            if (e + 2 < end) ref_next = sorted[e + 2];
        }
        if (more) xyzz30_madd_tail(acc, P, R);
    }
    // last run: [run_start, end)
    uint4* dst;
    if (run_start == b_beg && end == b_end) dst = buckets + (size_t)b * kXyzzU4;  // complete
    else if (run_start == start) dst = part_a + (size_t)lane * kXyzzU4;            // covers the whole segment
    else dst = part_b + (size_t)lane * kXyzzU4;                                    // tail shared with the next lane
    store_xyzz30(dst, acc);
}

// ---- table format -------------------------------------------------------------------------------------------------
// srs_kernels.hip builds the window table with the 12 x u32 field (x | y, 96 B of a 128-B record).  Once it is
// complete every record is rewritten in place into the accumulation kernel's native form: 13 signed radix-2^30
// digits of x * 2^390 reduced below 0.62 p, x in words 0..12 and y in words 16..28.  (0, 0) stays all zero.
__global__ void __launch_bounds__(256) k_table_to_fq(uint4* __restrict__ table, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    uint4* rec = table + i * kAffineU4;
    uint32_t w[24];
#pragma unroll
    for (int t = 0; t < 6; t++) {
        const uint4 v = rec[t];
        w[4 * t] = v.x; w[4 * t + 1] = v.y; w[4 * t + 2] = v.z; w[4 * t + 3] = v.w;
    }
    uint32_t any = 0;
#pragma unroll
    for (int t = 0; t < 24; t++) any |= w[t];
    uint32_t o[32];
#pragma unroll
    for (int t = 0; t < 32; t++) o[t] = 0;
    if (any) {
        const Fq x = fq_mul(fq_from_u32x12(w), fq_one());       // 64 s reduced: same residue, magnitude < 0.62 p
        const Fq y = fq_mul(fq_from_u32x12(w + 12), fq_one());
#pragma unroll
        for (int t = 0; t < kQ; t++) {
            o[t] = (uint32_t)x.d[t];
            o[16 + t] = (uint32_t)y.d[t];
        }
    }
#pragma unroll
    for (int t = 0; t < 8; t++) rec[t] = make_uint4(o[4 * t], o[4 * t + 1], o[4 * t + 2], o[4 * t + 3]);
}
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
void launch_table_to_fq(hipStream_t s, void* d_table, uint64_t records) {
    if (!records) return;
    hipLaunchKernelGGL(k_table_to_fq, dim3((unsigned)((records + 255) / 256)), dim3(256), 0, s, (uint4*)d_table, records);
}
void launch_affine_to_p1(hipStream_t s, const void* d_affine, uint32_t n, void* d_p1) {
    if (!n) return;
    hipLaunchKernelGGL(k_fq_table_to_p1, dim3((n + 63) / 64), dim3(64), 0, s, (const uint4*)d_affine, n, (uint4*)d_p1);
}

uint32_t accumulate_lanes(uint64_t max_refs, bool alone) {
    // 262144 segments = 1024 workgroups = two full rounds of the 512 that are resident at 256 VGPRs (two workgroups
    // per CU).  Measured at 2^20 terms with three slots in flight (round 2, signed radix-2^30 field): 262144 lanes
    // 2.60 ms per launch / 342 commitments/s, 196608 lanes 2.76 ms / 327, 131072 lanes 2.64 ms / 340.  (Round 1's
    // 206-VGPR kernel left room for the light kernels of the other slots beside it and preferred 196608 lanes in
    // company; this one fills the register file, so whatever is best alone is best in company too.)
    // KZG_ACCUM_LANES overrides for experiments.
    (void)alone;
    static const uint64_t target = [] {
        const char* v = std::getenv("KZG_ACCUM_LANES");
        uint64_t l = v ? std::strtoull(v, nullptr, 10) : 262144ull;
        return l < 64 ? 262144ull : (l > 262144ull ? 262144ull : l);
    }();
    const uint64_t lo = accumulate_min_seg(max_refs);
    uint64_t lanes = (max_refs + lo - 1) / lo;  // segments are at least 8 (tiny jobs: 4) references long
    if (lanes > target) lanes = target;
    lanes = (lanes + kAccumBlock - 1) / kAccumBlock * kAccumBlock;
    return (uint32_t)lanes;
}

void launch_bucket_accumulate(hipStream_t s, const void* d_table, const uint32_t* d_sorted, const uint32_t* d_offs,
                              uint32_t nb, uint32_t lanes, void* d_buckets, void* d_part_a, void* d_part_b,
                              uint32_t lds_reserve_bytes) {
    if (!lanes) return;
    // dynamic LDS: the prefetch slots (8 KiB per wave), or more when the caller reserves LDS to shape occupancy
    const uint32_t need = (kAccumBlock / 64) * 8 * 64 * 16;
    if (lds_reserve_bytes < need) lds_reserve_bytes = need;
    hipLaunchKernelGGL(k_bucket_accumulate, dim3(lanes / kAccumBlock), dim3(kAccumBlock), lds_reserve_bytes, s,
                       reinterpret_cast<const uint4*>(d_table), d_sorted, d_offs, nb, lanes,
                       reinterpret_cast<uint4*>(d_buckets), reinterpret_cast<uint4*>(d_part_a),
                       reinterpret_cast<uint4*>(d_part_b));
}

}  // namespace kzg
