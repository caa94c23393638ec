// poly_kernels.hip -- scalar-field synthetic division as a parallel suffix scan.
//
// Restates Polynomial::divide_by_root (reference src/polynomial.rs:150-195) on (P - y)
// (src/polynomial.rs:128-145): the reference's sequential recurrence  q[i-1] = c[i] + z q[i]
// is the suffix Horner scan
//        S[i] = sum_{k >= i} c[k] z^(k-i),     q[i-1] = S[i]  (i >= 1),     S[0] = P(z),
// and its final check  -z q[0] == c[0] - y  is  P(z) == y.  Subtracting y only changes c[0], which
// no S[i >= 1] reads, so the same scan serves Polynomial::evaluate (src/polynomial.rs:112-123).
//
// Two launches over chunks of L coefficients per lane (coefficients past n count as zero):
//   1. k_poly_chunks: chunk Horner values, a Kogge-Stone suffix scan inside the workgroup (multipliers z^(L 2^s), one
//      Fr product per step), workgroup aggregates A_u out;
//   2. k_poly_apply: every workgroup derives its own carry from the aggregates above it,
//      C_b = sum_{u > b} A_u zb^(u-b-1) with zb = z^(L*256): each lane takes the blocks b+1+t, b+1+t+256, ... (two
//      products per term, the powers from a 16 + 16 entry table), an addition tree over the lanes sums them -- no
//      workgroup waits for another and no third launch sits between the two; then every lane replays its chunk from
//      its now-known carry and writes q.  Workgroup 0 also leaves P(z) = A_0 + zb C_0.
//      (A version that let the last workgroup of launch 1 scan the aggregates behind an arrival counter measured 114 us
//      for that launch against 55: 512 agent-scope releases, each a write-back + invalidate of an XCD's L2.  Beyond 1024
//      blocks -- 2^21 coefficients -- a one-workgroup scan of the aggregates runs as a launch of its own.)
// Memory: a lane owns L CONSECUTIVE coefficients (the recurrence is sequential in the index), which read directly is a
// 256-byte lane stride: every wave-level load touches 64 lines for 16 bytes each, and the Horner loop waits for each
// one in turn (round 2: 52 + 22 + 35 us at 2^20, 2.4 x the algorithmic bytes at the memory controller).  Both kernels
// therefore move their tile through LDS: consecutive lanes load consecutive 16-byte words (all 16 loads of a lane in
// flight at once), the wave's 16 KiB land in LDS with one pad word per lane chunk (conflict-free both ways), each lane
// reads its own chunk from there; the quotient goes out the same way in reverse.
// Every power of z a lane needs (z^8 and its repeated squares for the scan inside a workgroup, z^2048 and the
// powers of the block stage, the 2 x 16 entry table of the replay) is computed ONCE on the host (~60 Fr products,
// host_fr.hpp) and travels as a kernel argument.
// Arithmetic: fr30.hip.h -- signed radix-2^30 digits, 162 multiply-adds per product instead of the ~500 instructions of
// the 8 x u32 carry chain.  The coefficients and every value that leaves the scan keep the ABI's form (x * 2^256, canonical
// 8 x u32); every multiplier is a power of z the host prepares as digits of w * 2^270 (fr30_host.hpp), so a product of a
// value and a multiplier is a value again and nothing is ever converted by multiplying.  Values inside the scan are lazy
// (a few r in magnitude, digits carry-normalised before each product); chunk values and aggregates travel between the two
// launches as digits (12 words), a result is made canonical where it leaves (fr30_to_limbs).
// Algorithmic bytes: 32 B read + 32 B written per coefficient; the coefficients are read
// twice (96 B per coefficient at the memory controller).  Bound by the Fr products at the occupancy 2^20 coefficients give.
#include "engine.h"
#include "fr30_host.hpp"

#include <cstring>

namespace kzg {

constexpr int kPolyL = 8;        // coefficients per lane.  Alone at 2^20: 4 -> 134 us, 8 -> 109, 16 -> 100 for the three kernels, but
                                 // with 16 the pipelined opening proofs drop 5 % (374-379 against 392-398 per s: half as many, longer
                                 // waves wait longer for room beside the other slots' accumulation), so 8 stays
constexpr int kPolyBlock = 256;  // lanes per workgroup
constexpr int kPolyTile = kPolyL * kPolyBlock;

constexpr int kPolyRec = 12;     // words of a value in digit form between the launches (9 digits, padded to 3 x 16 bytes)
size_t poly_chunk_words(uint32_t n) { return ((size_t)(n + kPolyTile - 1) / kPolyTile) * kPolyBlock * kPolyRec; }
size_t poly_block_words(uint32_t n) { return ((size_t)(n + kPolyTile - 1) / kPolyTile + 1) * kPolyRec; }

#define KZG_DEV __device__ __forceinline__

// a multiplier prepared by the host: digits of w * 2^270 (fr30_host.hpp)
struct FrArg {
    int32_t d[kR9];
};
KZG_DEV Fr30 fr_from_arg(const FrArg& a) {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < kR9; i++) r.d[i] = a.d[i];
    return r;
}
// value = value * multiplier + value (Horner step), carry-normalised
KZG_DEV Fr30 fr30_mul_add(const Fr30& h, const Fr30& mult, const Fr30& c) { return fr30_norm(fr30_add_raw(fr30_mul(h, mult), c)); }
// coefficient in the ABI's form -> digits
KZG_DEV Fr30 fr30_from_u4(const uint4& lo, const uint4& hi) {
    const uint32_t l[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return fr30_from_limbs(l);
}
KZG_DEV Fr30 load_coeff(const uint32_t* __restrict__ p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    return fr30_from_u4(q[0], q[1]);
}
// canonical 8 x u32 of a lazy value in (-r, 2r)
KZG_DEV void store_canonical(uint32_t* __restrict__ p, const Fr30& a) {
    uint32_t l[8];
    fr30_to_limbs(a, l);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(l[0], l[1], l[2], l[3]);
    q[1] = make_uint4(l[4], l[5], l[6], l[7]);
}
// digit records between the launches: one 12-word record per value (aggregates), or three planes of uint4 (chunk values:
// plane p of lane t at [p * lanes + t], consecutive lanes on consecutive words)
KZG_DEV Fr30 load_rec(const uint32_t* __restrict__ p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2];
    Fr30 r;
    r.d[0] = (int32_t)a.x; r.d[1] = (int32_t)a.y; r.d[2] = (int32_t)a.z; r.d[3] = (int32_t)a.w;
    r.d[4] = (int32_t)b.x; r.d[5] = (int32_t)b.y; r.d[6] = (int32_t)b.z; r.d[7] = (int32_t)b.w;
    r.d[8] = (int32_t)c.x;
    return r;
}
KZG_DEV void store_rec(uint32_t* __restrict__ p, const Fr30& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4((uint32_t)a.d[0], (uint32_t)a.d[1], (uint32_t)a.d[2], (uint32_t)a.d[3]);
    q[1] = make_uint4((uint32_t)a.d[4], (uint32_t)a.d[5], (uint32_t)a.d[6], (uint32_t)a.d[7]);
    q[2] = make_uint4((uint32_t)a.d[8], 0u, 0u, 0u);
}
KZG_DEV Fr30 load_planes(const uint32_t* __restrict__ base, size_t lanes, size_t t) {
    const uint4* q = reinterpret_cast<const uint4*>(base);
    const uint4 a = q[t], b = q[lanes + t], c = q[2 * lanes + t];
    Fr30 r;
    r.d[0] = (int32_t)a.x; r.d[1] = (int32_t)a.y; r.d[2] = (int32_t)a.z; r.d[3] = (int32_t)a.w;
    r.d[4] = (int32_t)b.x; r.d[5] = (int32_t)b.y; r.d[6] = (int32_t)b.z; r.d[7] = (int32_t)b.w;
    r.d[8] = (int32_t)c.x;
    return r;
}
KZG_DEV void store_planes(uint32_t* __restrict__ base, size_t lanes, size_t t, const Fr30& a) {
    uint4* q = reinterpret_cast<uint4*>(base);
    q[t] = make_uint4((uint32_t)a.d[0], (uint32_t)a.d[1], (uint32_t)a.d[2], (uint32_t)a.d[3]);
    q[lanes + t] = make_uint4((uint32_t)a.d[4], (uint32_t)a.d[5], (uint32_t)a.d[6], (uint32_t)a.d[7]);
    q[2 * lanes + t] = make_uint4((uint32_t)a.d[8], 0u, 0u, 0u);
}
// powers of z prepared on the host (launch_quotient)
struct PolyPowers {
    FrArg z;
    FrArg one;         // 1: a product by it brings a lazy value back under r / 2 in magnitude
    FrArg zl_sq[8];    // (z^L)^(2^s): multipliers of the scan inside a workgroup
    FrArg zb;          // z^(L * 256): one workgroup up
    FrArg zbp_sq[8];   // (zb^per)^(2^s): multipliers of the scan over the lanes of the block stage (more than kPolyDirectBlocks blocks)
    FrArg ba[16];      // zb^(16 e)
    FrArg bb[16];      // zb^e
    FrArg zb256;       // zb^256
    FrArg pa[16];      // (z^L)^(16 e)
    FrArg pb[16];      // (z^L)^e
};
constexpr int kPolyScanWords = kPolyBlock * kR9;  // LDS words of the exchanges below
// Kogge-Stone suffix scan over the workgroup: v_t <- sum_{u >= t} v_u * mult^(u - t)
template <int BLOCK>
KZG_DEV Fr30 block_suffix_scan(Fr30 v, const FrArg* mults /* mult^(2^s), s < log2(BLOCK) */, uint32_t* lds /* BLOCK * 9 words */) {
    const int t = threadIdx.x;
    int s = 0;
    for (int off = 1; off < BLOCK; off <<= 1, s++) {
        const Fr30 mult = fr_from_arg(mults[s]);
#pragma unroll
        for (int i = 0; i < kR9; i++) lds[i * BLOCK + t] = (uint32_t)v.d[i];
        __syncthreads();
        if (t + off < BLOCK) {
            Fr30 o;
#pragma unroll
            for (int i = 0; i < kR9; i++) o.d[i] = (int32_t)lds[i * BLOCK + t + off];
            v = fr30_mul_add(o, mult, v);
        }
        __syncthreads();
    }
    return v;
}

// ---- tile staging through LDS -------------------------------------------------------------------------------------
// A wave owns 64 x kPolyL coefficients = 64 x 16 uint4.  Global side: word g = i * 64 + lane (i < 16), consecutive lanes
// on consecutive 16-byte words.  LDS side: word g sits at g + g / 16 (one pad word behind every lane chunk of 16), so the
// row-wise writes are contiguous runs of 16 and the chunk-wise reads of lane l (words 17 l + j) fall on different banks
// for 16 consecutive lanes.  Only the lanes of one wave touch one region: wave-level ordering is enough.
constexpr int kPolyWaveWords = 64 * 2 * kPolyL;                         // uint4 per wave (1024)
constexpr int kPolyWaveLds = kPolyWaveWords + kPolyWaveWords / 16;      // padded (1088)
static_assert(kPolyL == 8, "the staging below is laid out for 16 uint4 per lane");
KZG_DEV uint32_t poly_lds_slot(uint32_t g) { return g + (g >> 4); }
KZG_DEV void poly_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// coefficients [first, first + 512) of `coeffs` (zero past n) -> the wave's LDS region
KZG_DEV void poly_stage_in(const uint32_t* __restrict__ coeffs, uint64_t first, uint32_t n, uint4* __restrict__ lds_wave, uint32_t lane) {
    const uint4* src = reinterpret_cast<const uint4*>(coeffs) + 2 * first;
    const uint64_t valid = first < n ? 2 * ((uint64_t)n - first) : 0;  // uint4 words of the region that exist
    // every load is issued unconditionally (words past the end read coefficient 0 instead and are zeroed afterwards): a
    // guarded load becomes a branch with a wait behind it, and the 16 loads of a lane must be in flight together
    uint4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t g = (uint32_t)i * 64u + lane;
        const uint4* p = g < valid ? src + g : reinterpret_cast<const uint4*>(coeffs);
        v[i] = *p;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t g = (uint32_t)i * 64u + lane;
        lds_wave[poly_lds_slot(g)] = g < valid ? v[i] : make_uint4(0, 0, 0, 0);
    }
    poly_wave_sync();
}
KZG_DEV Fr30 poly_lds_coeff(const uint4* __restrict__ lds_wave, uint32_t lane, int k, bool* nonzero = nullptr) {
    const uint4 lo = lds_wave[17u * lane + 2u * (uint32_t)k], hi = lds_wave[17u * lane + 2u * (uint32_t)k + 1u];
    if (nonzero) *nonzero = ((lo.x | lo.y | lo.z | lo.w) | (hi.x | hi.y | hi.z | hi.w)) != 0u;
    return fr30_from_u4(lo, hi);
}

// single workgroup's worth of work: carries for every block.  d_block[b] in: aggregate of block b (zero carry-in);
// out: d_block[b] = S at the first coefficient of block b+1 (its carry-in); d_result = P(z).
// Lane t owns `per` consecutive blocks: Horner over its blocks, Kogge-Stone across lanes, replay.
KZG_DEV void poly_block_stage(uint32_t* __restrict__ d_block, uint32_t nblocks, const PolyPowers& pw, uint32_t* __restrict__ d_result,
                              uint32_t* lds) {
    const Fr30 zb = fr_from_arg(pw.zb);  // one block up
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nblocks + kPolyBlock - 1) / kPolyBlock;
    const uint32_t lo = t * per < nblocks ? t * per : nblocks;
    const uint32_t hi = lo + per < nblocks ? lo + per : nblocks;
    Fr30 h = fr30_zero();
    for (uint32_t u = hi; u-- > lo;) h = fr30_mul_add(h, zb, load_rec(d_block + (size_t)u * kPolyRec));
    // H_t = sum_{v >= t} h_v * (zb^per)^(v - t)
    Fr30 H = block_suffix_scan<kPolyBlock>(h, pw.zbp_sq, lds);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kR9; i++) lds[i * kPolyBlock + t] = (uint32_t)H.d[i];
    __syncthreads();
    Fr30 s = fr30_zero();  // S at the first coefficient of the next lane's first block
    if (t + 1 < (uint32_t)kPolyBlock) {
#pragma unroll
        for (int i = 0; i < kR9; i++) s.d[i] = (int32_t)lds[i * kPolyBlock + t + 1];
    }
    for (uint32_t u = hi; u-- > lo;) {
        const Fr30 a = load_rec(d_block + (size_t)u * kPolyRec);
        store_rec(d_block + (size_t)u * kPolyRec, s);  // carry-in of block u = S at the start of block u+1
        s = fr30_mul_add(s, zb, a);
    }
    if (t == 0) store_canonical(d_result, fr30_mul(s, fr_from_arg(pw.one)));  // S[0] = P(z)
}

// d_flags: the job's flag words -- [0] |= any non-zero coefficient with index >= 1, [16..23] receive c[0] (what the
// constant-polynomial rule compares with y, src/polynomial.rs:159-167).
__global__ void __launch_bounds__(kPolyBlock) k_poly_chunks(const uint32_t* __restrict__ coeffs, uint32_t n, PolyPowers pw,
                                                            uint32_t* __restrict__ d_chunk,
                                                            uint32_t* __restrict__ d_block,
                                                            uint32_t* __restrict__ d_flags) {
    extern __shared__ uint4 lds_tile[];                       // 4 waves x kPolyWaveLds uint4
    __shared__ uint32_t lds[kPolyScanWords];
    const Fr30 z = fr_from_arg(pw.z);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4* lds_wave = lds_tile + wave * kPolyWaveLds;
    const uint32_t t = blockIdx.x * kPolyBlock + threadIdx.x;
    const uint64_t wave_first = ((uint64_t)blockIdx.x * kPolyBlock + wave * 64u) * kPolyL;
    poly_stage_in(coeffs, wave_first, n, lds_wave, lane);
    Fr30 h = fr30_zero();
    bool nz = false;
#pragma unroll
    for (int k = kPolyL - 1; k >= 0; k--) {
        bool c_nz;
        const Fr30 c = poly_lds_coeff(lds_wave, lane, k, &c_nz);
        if (((uint64_t)t * kPolyL + k) >= 1 && c_nz) nz = true;  // (coefficients past n were staged as zero)
        h = fr30_mul_add(h, z, c);
    }
    if (__any(nz) && lane == 0) d_flags[0] = 1u;  // (the flag words are host-mapped: plain stores, every writer stores 1)
    if (t == 0) {  // c[0], as it came
        reinterpret_cast<uint4*>(d_flags + 16)[0] = lds_wave[0];
        reinterpret_cast<uint4*>(d_flags + 16)[1] = lds_wave[1];
    }
    h = block_suffix_scan<kPolyBlock>(h, pw.zl_sq, lds);
    store_planes(d_chunk, (size_t)gridDim.x * kPolyBlock, t, h);
    if (threadIdx.x == 0) store_rec(d_block + (size_t)blockIdx.x * kPolyRec, h);
}

__global__ void __launch_bounds__(kPolyBlock) k_poly_blocks(uint32_t* __restrict__ d_block, uint32_t nblocks, PolyPowers pw,
                                                            uint32_t* __restrict__ d_result) {
    __shared__ uint32_t lds[kPolyScanWords];
    poly_block_stage(d_block, nblocks, pw, d_result, lds);
}

// C_b = sum_{u > b} A_u zb^(u-b-1) for this workgroup's block b (b = -1: everything, i.e. P(z)): terms spread over the
// lanes, summed by an addition tree through LDS; every lane returns the sum
constexpr uint32_t kPolyDirectBlocks = 1024;  // beyond: k_poly_blocks prepares the carries
KZG_DEV Fr30 poly_carry_from_aggregates(const uint32_t* __restrict__ d_block, uint32_t nblocks, uint32_t first /* b + 1 */,
                                        const PolyPowers& pw, uint32_t* lds /* kPolyBlock * 9 words */) {
    const uint32_t t = threadIdx.x;
    Fr30 sum = fr30_zero();
    if (first + t < nblocks) {
        Fr30 power = fr30_mul(fr_from_arg(pw.ba[t >> 4]), fr_from_arg(pw.bb[t & 15]));  // zb^t (a multiplier again)
        const Fr30 stride = fr_from_arg(pw.zb256);
        for (uint32_t u = first + t; u < nblocks; u += kPolyBlock) {
            sum = fr30_mul_add(load_rec(d_block + (size_t)u * kPolyRec), power, sum);
            if (u + kPolyBlock < nblocks) power = fr30_mul(power, stride);
        }
    }
    for (int off = kPolyBlock / 2; off >= 1; off >>= 1) {
#pragma unroll
        for (int i = 0; i < kR9; i++) lds[i * kPolyBlock + t] = (uint32_t)sum.d[i];
        __syncthreads();
        if ((int)t < off) {
            Fr30 o;
#pragma unroll
            for (int i = 0; i < kR9; i++) o.d[i] = (int32_t)lds[i * kPolyBlock + t + off];
            sum = fr30_add(sum, o);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < kR9; i++) lds[i * kPolyBlock + t] = (uint32_t)sum.d[i];
    __syncthreads();
    Fr30 total;
#pragma unroll
    for (int i = 0; i < kR9; i++) total.d[i] = (int32_t)lds[i * kPolyBlock];
    return total;
}

// direct != 0: d_block holds the aggregates A_u (carries are derived here, workgroup 0 writes P(z) to d_result);
// direct == 0: d_block holds the carries (k_poly_blocks ran).  d_q == nullptr: only P(z) is wanted (one workgroup).
__global__ void __launch_bounds__(kPolyBlock) k_poly_apply(const uint32_t* __restrict__ coeffs, uint32_t n, PolyPowers pw,
                                                           const uint32_t* __restrict__ d_chunk,
                                                           const uint32_t* __restrict__ d_block, uint32_t nblocks, int direct,
                                                           uint32_t* __restrict__ d_q, uint32_t* __restrict__ d_result) {
    extern __shared__ uint4 lds_tile[];
    __shared__ uint32_t lds[kPolyScanWords];
    const Fr30 z = fr_from_arg(pw.z);
    const int tl = threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint4* lds_wave = lds_tile + wave * kPolyWaveLds;
    const uint32_t t = blockIdx.x * kPolyBlock + tl;
    const uint64_t wave_first = ((uint64_t)blockIdx.x * kPolyBlock + wave * 64u) * kPolyL;
    if (d_q) poly_stage_in(coeffs, wave_first, n, lds_wave, lane);  // in flight while the carry is worked out
    // carry into this workgroup's block = S at the first coefficient of the next block
    Fr30 blk_carry;
    if (direct) {
        blk_carry = poly_carry_from_aggregates(d_block, nblocks, blockIdx.x + 1, pw, lds);
        if (blockIdx.x == 0 && tl == 0) {  // P(z) = S[0] = A_0 + zb C_0, brought under r / 2 by a product with one
            const Fr30 s0 = fr30_mul_add(blk_carry, fr_from_arg(pw.zb), load_rec(d_block));
            store_canonical(d_result, fr30_mul(s0, fr_from_arg(pw.one)));
        }
    } else {
        blk_carry = load_rec(d_block + (size_t)blockIdx.x * kPolyRec);
    }
    if (!d_q) return;
    Fr30 h;
    uint32_t dist = (uint32_t)(kPolyBlock - 1 - tl);  // chunks between the next chunk and the block end
    const Fr30 pa = fr_from_arg(pw.pa[dist >> 4]), pb = fr_from_arg(pw.pb[dist & 15]);  // (z^L)^dist = pa * pb
    const Fr30 scaled = fr30_mul(blk_carry, fr30_mul(pa, pb));
    if (tl + 1 < kPolyBlock) {
        h = fr30_add(load_planes(d_chunk, (size_t)gridDim.x * kPolyBlock, (size_t)t + 1), scaled);
    } else {
        h = scaled;  // dist == 0: the block carry itself
    }
    // replay: S at every coefficient of the chunk (a product plus a canonical coefficient: inside (-r, 2r)), written back
    // canonical into the lane's own LDS words
#pragma unroll
    for (int k = kPolyL - 1; k >= 0; k--) {
        h = fr30_mul_add(h, z, poly_lds_coeff(lds_wave, lane, k));
        uint32_t l[8];
        fr30_to_limbs(h, l);
        lds_wave[17u * lane + 2u * (uint32_t)k] = make_uint4(l[0], l[1], l[2], l[3]);
        lds_wave[17u * lane + 2u * (uint32_t)k + 1u] = make_uint4(l[4], l[5], l[6], l[7]);
    }
    poly_wave_sync();
    // q[i - 1] = S[i]: the wave's 512 values go out one coefficient lower, consecutive lanes on consecutive words
    uint4* dst = reinterpret_cast<uint4*>(d_q) + 2 * wave_first;  // word g of the region belongs at dst[g - 2]
    const uint64_t valid = wave_first < n ? 2 * ((uint64_t)n - wave_first) : 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t g = (uint32_t)i * 64u + lane;
        if (g < valid && (wave_first != 0 || g >= 2)) dst[(int64_t)g - 2] = lds_wave[poly_lds_slot(g)];
    }
}

// Up to kPolySingleMax coefficients (the reference's bench degrees): the whole scan in ONE workgroup and one launch --
// chunk Horner, Kogge-Stone over the 256 lanes, replay from the neighbour's value.  It also writes the 64 flag words
// of the slot in full (P(z) at [8..15], c[0] at [16..23], [0] = any non-zero coefficient with index >= 1, zero
// elsewhere), which saves the caller a memset and a device-to-device copy: five stream operations become one.
constexpr uint32_t kPolySingleL = 16;
constexpr uint32_t kPolySingleMax = kPolySingleL * kPolyBlock;  // 4096
__global__ void __launch_bounds__(kPolyBlock) k_poly_single(const uint32_t* __restrict__ coeffs, uint32_t n, PolyPowers pw,
                                                            uint32_t* __restrict__ d_q, uint32_t* __restrict__ d_small) {
    __shared__ uint32_t lds[kPolyScanWords];
    const Fr30 z = fr_from_arg(pw.z);
    const uint32_t t = threadIdx.x;
    const uint32_t base = t * kPolySingleL;
    Fr30 h = fr30_zero();
    bool nz = false;
#pragma unroll 1
    for (int k = (int)kPolySingleL - 1; k >= 0; k--) {
        const uint32_t idx = base + k;
        Fr30 c = fr30_zero();
        if (idx < n) {
            const uint4* q = reinterpret_cast<const uint4*>(coeffs + (size_t)idx * 8);
            const uint4 lo = q[0], hi = q[1];
            if (idx >= 1 && ((lo.x | lo.y | lo.z | lo.w) | (hi.x | hi.y | hi.z | hi.w)) != 0u) nz = true;
            c = fr30_from_u4(lo, hi);
        }
        h = fr30_mul_add(h, z, c);
    }
    const int any_nz = __syncthreads_or(nz ? 1 : 0);
    h = block_suffix_scan<kPolyBlock>(h, pw.zl_sq, lds);  // S at the first coefficient of this lane's chunk
#pragma unroll
    for (int i = 0; i < kR9; i++) lds[i * kPolyBlock + t] = (uint32_t)h.d[i];
    __syncthreads();
    if (t < 64) d_small[t] = 0;
    __syncthreads();
    if (t == 0) {
        d_small[0] = any_nz ? 1u : 0u;
        store_canonical(d_small + 8, fr30_mul(h, fr_from_arg(pw.one)));  // S[0] = P(z)
        const uint4* q = reinterpret_cast<const uint4*>(coeffs);
        reinterpret_cast<uint4*>(d_small + 16)[0] = q[0];
        reinterpret_cast<uint4*>(d_small + 16)[1] = q[1];
    }
    if (!d_q || n <= 1) return;
    Fr30 carry = fr30_zero();  // S at the first coefficient of the next chunk
    if (t + 1 < (uint32_t)kPolyBlock) {
#pragma unroll
        for (int i = 0; i < kR9; i++) carry.d[i] = (int32_t)lds[i * kPolyBlock + t + 1];
    }
    h = carry;
#pragma unroll 1
    for (int k = (int)kPolySingleL - 1; k >= 0; k--) {
        const uint32_t idx = base + k;
        if (idx < n) {
            h = fr30_mul_add(h, z, load_coeff(coeffs + (size_t)idx * 8));  // a product plus a canonical coefficient
            if (idx >= 1) store_canonical(d_q + (size_t)(idx - 1) * 8, h);
        } else {
            h = fr30_mul(h, z);  // (past the end: h is zero and stays zero)
        }
    }
}

// the scan kernels stage a tile of 68 KiB: more than the 64 KiB a workgroup gets by default (once per device)
constexpr uint32_t kPolyTileLds = (kPolyBlock / 64) * kPolyWaveLds * 16;
bool poly_prepare_device() {
    const bool a = hipFuncSetAttribute((const void*)k_poly_chunks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPolyTileLds) == hipSuccess;
    const bool b = hipFuncSetAttribute((const void*)k_poly_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPolyTileLds) == hipSuccess;
    if (!(a && b)) (void)hipGetLastError();
    return a && b;
}

bool launch_quotient_single(hipStream_t s, const uint32_t* d_coeffs, uint32_t n, const uint32_t z_mont[8], uint32_t* d_q,
                            uint32_t* d_small) {
    if (n == 0 || n > kPolySingleMax) return false;
    namespace hf = kzg_host;
    PolyPowers pw;
    std::memset(&pw, 0, sizeof pw);
    auto put = [](FrArg& dst, const hf::Fr& v) {  // the multiplier's form: digits of v * 2^270
        const Fr30 d = fr30_arg_from_mont256(v);
        std::memcpy(dst.d, d.d, sizeof dst.d);
    };
    hf::Fr z;
    std::memcpy(z.l, z_mont, 32);
    put(pw.z, z);
    put(pw.one, hf::kFrOne);
    hf::Fr m = hf::fr_pow(z, kPolySingleL);
    for (int k = 0; k < 8; k++) {  // (z^L)^(2^k)
        put(pw.zl_sq[k], m);
        m = hf::fr_mul(m, m);
    }
    hipLaunchKernelGGL(k_poly_single, dim3(1), dim3(kPolyBlock), 0, s, d_coeffs, n, pw, d_q, d_small);
    return true;
}

void launch_quotient(hipStream_t s, const uint32_t* d_coeffs, uint32_t n, const uint32_t z_mont[8], uint32_t* d_q,
                     PolyScratch sc) {
    if (n == 0) return;
    namespace hf = kzg_host;
    const uint32_t nblocks = (n + kPolyTile - 1) / kPolyTile;
    const uint32_t per = (nblocks + kPolyBlock - 1) / kPolyBlock;  // blocks per lane of the block stage
    PolyPowers pw;
    auto put = [](FrArg& dst, const hf::Fr& v) {  // the multiplier's form: digits of v * 2^270
        const Fr30 d = fr30_arg_from_mont256(v);
        std::memcpy(dst.d, d.d, sizeof dst.d);
    };
    hf::Fr z;
    std::memcpy(z.l, z_mont, 32);
    put(pw.z, z);
    put(pw.one, hf::kFrOne);
    hf::Fr zl = hf::fr_pow(z, kPolyL), m = zl;
    for (int k = 0; k < 8; k++) {  // zl^(2^k); after the loop m = zl^256 = z^(L * 256)
        put(pw.zl_sq[k], m);
        m = hf::fr_mul(m, m);
    }
    put(pw.zb, m);
    hf::Fr mb = hf::fr_pow(m, per);
    for (int k = 0; k < 8; k++) {
        put(pw.zbp_sq[k], mb);
        mb = hf::fr_mul(mb, mb);
    }
    hf::Fr zl16 = hf::fr_pow(zl, 16), a = hf::kFrOne, b = hf::kFrOne;
    for (int e = 0; e < 16; e++) {
        put(pw.pa[e], a);
        put(pw.pb[e], b);
        a = hf::fr_mul(a, zl16);
        b = hf::fr_mul(b, zl);
    }
    hf::Fr m16 = hf::fr_pow(m, 16), ba = hf::kFrOne, bb = hf::kFrOne;
    for (int e = 0; e < 16; e++) {
        put(pw.ba[e], ba);
        put(pw.bb[e], bb);
        ba = hf::fr_mul(ba, m16);
        bb = hf::fr_mul(bb, m);
    }
    put(pw.zb256, ba);  // (zb^16)^16
    constexpr uint32_t tile_lds = kPolyTileLds;
    hipLaunchKernelGGL(k_poly_chunks, dim3(nblocks), dim3(kPolyBlock), tile_lds, s, d_coeffs, n, pw, sc.d_chunk, sc.d_block, sc.d_flags);
    const int direct = nblocks <= kPolyDirectBlocks;
    if (!direct) hipLaunchKernelGGL(k_poly_blocks, dim3(1), dim3(kPolyBlock), 0, s, sc.d_block, nblocks, pw, sc.d_result);
    uint32_t* q = (d_q && n > 1) ? d_q : nullptr;
    if (q || direct)  // (without a quotient: one workgroup that leaves P(z))
        hipLaunchKernelGGL(k_poly_apply, dim3(q ? nblocks : 1), dim3(kPolyBlock), q ? tile_lds : 0u, s, d_coeffs, n, pw, sc.d_chunk,
                           sc.d_block, nblocks, direct, q, sc.d_result);
}

}  // namespace kzg
