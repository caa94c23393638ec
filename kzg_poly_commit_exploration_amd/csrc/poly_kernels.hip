// poly_kernels.hip -- scalar-field synthetic division as a parallel suffix scan.
//
// Restates Polynomial::divide_by_root (reference src/polynomial.rs:150-195) on (P - y)
// (src/polynomial.rs:128-145): the reference's sequential recurrence  q[i-1] = c[i] + z q[i]
// is the suffix Horner scan
//        S[i] = sum_{k >= i} c[k] z^(k-i),     q[i-1] = S[i]  (i >= 1),     S[0] = P(z),
// and its final check  -z q[0] == c[0] - y  is  P(z) == y.  Subtracting y only changes c[0], which
// no S[i >= 1] reads, so the same scan serves Polynomial::evaluate (src/polynomial.rs:112-123).
//
// Three launches over chunks of L coefficients per lane (coefficients past n count as zero):
//   1. chunk Horner values, then a Kogge-Stone suffix scan inside the workgroup (multipliers
//      z^(L 2^s), one Fr product per step), workgroup aggregates out;
//   2. one workgroup scans the aggregates (multiplier z^(L*256); each lane takes a run of blocks);
//   3. every lane replays its chunk from its now-known carry and writes q.
// Every power of z a lane needs (z^8 and its repeated squares for the scan inside a workgroup, z^2048 and the
// powers of the block stage, the 2 x 16 entry table of the replay) is computed ONCE on the host (~60 Fr products,
// host_fr.hpp) and travels as a kernel argument: the per-lane chains of dependent Fr products drop from 28 / 31 / 22
// to 16 / 12 / 10 in the three kernels, which is what these latency-bound launches cost.
// Algorithmic bytes: 32 B read + 32 B written per coefficient; the chunk
// is read twice (the second time mostly from L2 / Infinity Cache).  HBM / latency bound.
#include "engine.h"
#include "field.hip.h"
#include "host_fr.hpp"

#include <cstring>

namespace kzg {

constexpr int kPolyL = 8;        // coefficients per lane.  Alone at 2^20: 4 -> 134 us, 8 -> 109, 16 -> 100 for the three kernels, but
                                 // with 16 the pipelined opening proofs drop 5 % (374-379 against 392-398 per s: half as many, longer
                                 // waves wait longer for room beside the other slots' accumulation), so 8 stays
constexpr int kPolyBlock = 256;  // lanes per workgroup
constexpr int kPolyTile = kPolyL * kPolyBlock;

size_t poly_chunk_words(uint32_t n) { return ((size_t)(n + kPolyTile - 1) / kPolyTile) * kPolyBlock * 8; }
size_t poly_block_words(uint32_t n) { return ((size_t)(n + kPolyTile - 1) / kPolyTile + 1) * 8; }

KZG_DEV Fr load_fr(const uint32_t* __restrict__ p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    Fr a;
    a.l[0] = lo.x; a.l[1] = lo.y; a.l[2] = lo.z; a.l[3] = lo.w;
    a.l[4] = hi.x; a.l[5] = hi.y; a.l[6] = hi.z; a.l[7] = hi.w;
    return a;
}
KZG_DEV void store_fr(uint32_t* __restrict__ p, const Fr& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
    q[1] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
}
struct FrArg {
    uint32_t l[8];
};
KZG_DEV Fr fr_from_arg(const FrArg& a) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = a.l[i];
    return r;
}
// powers of z prepared on the host (launch_quotient)
struct PolyPowers {
    FrArg z;
    FrArg zl_sq[8];    // (z^L)^(2^s): multipliers of the scan inside a workgroup
    FrArg zb;          // z^(L * 256): one workgroup up
    FrArg zbp_sq[8];   // (zb^per)^(2^s): multipliers of the scan over the lanes of the block stage
    FrArg pa[16];      // (z^L)^(16 e)
    FrArg pb[16];      // (z^L)^e
};
// Kogge-Stone suffix scan over the workgroup: v_t <- sum_{u >= t} v_u * mult^(u - t)
template <int BLOCK>
KZG_DEV Fr block_suffix_scan(Fr v, const FrArg* mults /* mult^(2^s), s < log2(BLOCK) */, uint32_t* lds /* BLOCK * 8 words */) {
    const int t = threadIdx.x;
    int s = 0;
    for (int off = 1; off < BLOCK; off <<= 1, s++) {
        const Fr mult = fr_from_arg(mults[s]);
#pragma unroll
        for (int i = 0; i < 8; i++) lds[i * BLOCK + t] = v.l[i];
        __syncthreads();
        if (t + off < BLOCK) {
            Fr o;
#pragma unroll
            for (int i = 0; i < 8; i++) o.l[i] = lds[i * BLOCK + t + off];
            v = fe_add(v, fe_mul(mult, o));
        }
        __syncthreads();
    }
    return v;
}

__global__ void __launch_bounds__(kPolyBlock) k_poly_chunks(const uint32_t* __restrict__ coeffs, uint32_t n, PolyPowers pw,
                                                            uint32_t* __restrict__ d_chunk,
                                                            uint32_t* __restrict__ d_block,
                                                            uint32_t* __restrict__ d_flags) {
    __shared__ uint32_t lds[kPolyBlock * 8];
    const Fr z = fr_from_arg(pw.z);
    const uint32_t t = blockIdx.x * kPolyBlock + threadIdx.x;
    const uint32_t base = t * kPolyL;
    Fr h = Fr::zero();
    bool nz = false;
#pragma unroll 1
    for (int k = kPolyL - 1; k >= 0; k--) {
        uint32_t idx = base + k;
        h = fe_mul(h, z);
        if (idx < n) {
            Fr c = load_fr(coeffs + (size_t)idx * 8);
            if (idx >= 1 && !c.is_zero()) nz = true;
            h = fe_add(h, c);
        }
    }
    if (__any(nz) && (threadIdx.x & 63) == 0) atomicOr(&d_flags[0], 1u);
    h = block_suffix_scan<kPolyBlock>(h, pw.zl_sq, lds);
    store_fr(d_chunk + (size_t)t * 8, h);
    if (threadIdx.x == 0) store_fr(d_block + (size_t)blockIdx.x * 8, h);
}

// single workgroup: carries for every block.  d_block[b] in: aggregate of block b (zero carry-in);
// out: d_block[b] = S at the first coefficient of block b+1 (its carry-in); d_result = P(z).
// 256 lanes (one wave per SIMD, 86 VGPRs): small enough to start beside two resident accumulation waves of
// another slot -- the former 1024-lane version needed 4 x 86 VGPRs per SIMD and waited ~1 ms for them.
// Lane t owns `per` consecutive blocks: Horner over its blocks, Kogge-Stone across lanes, replay.
__global__ void __launch_bounds__(kPolyBlock) k_poly_blocks(uint32_t* __restrict__ d_block, uint32_t nblocks, PolyPowers pw,
                                                            uint32_t* __restrict__ d_result) {
    __shared__ uint32_t lds[kPolyBlock * 8];
    const Fr zb = fr_from_arg(pw.zb);  // one block up
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nblocks + kPolyBlock - 1) / kPolyBlock;
    const uint32_t lo = t * per;
    const uint32_t hi = lo + per < nblocks ? lo + per : nblocks;
    Fr h = Fr::zero();
    for (uint32_t u = hi; u-- > lo;) h = fe_add(fe_mul(h, zb), load_fr(d_block + (size_t)u * 8));
    // H_t = sum_{v >= t} h_v * (zb^per)^(v - t)
    Fr H = block_suffix_scan<kPolyBlock>(h, pw.zbp_sq, lds);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) lds[i * kPolyBlock + t] = H.l[i];
    __syncthreads();
    Fr s = Fr::zero();  // S at the first coefficient of the next lane's first block
    if (t + 1 < (uint32_t)kPolyBlock) {
#pragma unroll
        for (int i = 0; i < 8; i++) s.l[i] = lds[i * kPolyBlock + t + 1];
    }
    for (uint32_t u = hi; u-- > lo;) {
        Fr a = load_fr(d_block + (size_t)u * 8);
        store_fr(d_block + (size_t)u * 8, s);  // carry-in of block u = S at the start of block u+1
        s = fe_add(fe_mul(s, zb), a);
    }
    if (t == 0) store_fr(d_result, s);  // S[0] = P(z)
}

__global__ void __launch_bounds__(kPolyBlock) k_poly_apply(const uint32_t* __restrict__ coeffs, uint32_t n, PolyPowers pw,
                                                           const uint32_t* __restrict__ d_chunk,
                                                           const uint32_t* __restrict__ d_block,
                                                           uint32_t* __restrict__ d_q) {
    const Fr z = fr_from_arg(pw.z);
    const int tl = threadIdx.x;
    const uint32_t t = blockIdx.x * kPolyBlock + tl;
    // carry into this lane's chunk = S at the first coefficient of the next chunk
    Fr blk_carry = load_fr(d_block + (size_t)blockIdx.x * 8);
    Fr h;
    uint32_t dist = (uint32_t)(kPolyBlock - 1 - tl);  // chunks between the next chunk and the block end
    const Fr pa = fr_from_arg(pw.pa[dist >> 4]), pb = fr_from_arg(pw.pb[dist & 15]);  // (z^L)^dist = pa * pb
    Fr scaled = fe_mul(fe_mul(pa, pb), blk_carry);
    if (tl + 1 < kPolyBlock) {
        h = fe_add(load_fr(d_chunk + (size_t)(t + 1) * 8), scaled);
    } else {
        h = scaled;  // dist == 0: the block carry itself
    }
    const uint32_t base = t * kPolyL;
#pragma unroll 1
    for (int k = kPolyL - 1; k >= 0; k--) {
        uint32_t idx = base + k;
        h = fe_mul(h, z);
        if (idx < n) {
            h = fe_add(h, load_fr(coeffs + (size_t)idx * 8));
            if (idx >= 1) store_fr(d_q + (size_t)(idx - 1) * 8, h);
        }
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
    __shared__ uint32_t lds[kPolyBlock * 8];
    const Fr z = fr_from_arg(pw.z);
    const uint32_t t = threadIdx.x;
    const uint32_t base = t * kPolySingleL;
    Fr h = Fr::zero();
    bool nz = false;
#pragma unroll 1
    for (int k = (int)kPolySingleL - 1; k >= 0; k--) {
        const uint32_t idx = base + k;
        h = fe_mul(h, z);
        if (idx < n) {
            const Fr c = load_fr(coeffs + (size_t)idx * 8);
            if (idx >= 1 && !c.is_zero()) nz = true;
            h = fe_add(h, c);
        }
    }
    const int any_nz = __syncthreads_or(nz ? 1 : 0);
    h = block_suffix_scan<kPolyBlock>(h, pw.zl_sq, lds);  // S at the first coefficient of this lane's chunk
#pragma unroll
    for (int i = 0; i < 8; i++) lds[i * kPolyBlock + t] = h.l[i];
    __syncthreads();
    if (t < 64) d_small[t] = 0;
    __syncthreads();
    if (t == 0) {
        d_small[0] = any_nz ? 1u : 0u;
        store_fr(d_small + 8, h);  // S[0] = P(z)
        store_fr(d_small + 16, load_fr(coeffs));
    }
    if (!d_q || n <= 1) return;
    Fr carry = Fr::zero();  // S at the first coefficient of the next chunk
    if (t + 1 < (uint32_t)kPolyBlock) {
#pragma unroll
        for (int i = 0; i < 8; i++) carry.l[i] = lds[i * kPolyBlock + t + 1];
    }
    h = carry;
#pragma unroll 1
    for (int k = (int)kPolySingleL - 1; k >= 0; k--) {
        const uint32_t idx = base + k;
        h = fe_mul(h, z);
        if (idx < n) {
            h = fe_add(h, load_fr(coeffs + (size_t)idx * 8));
            if (idx >= 1) store_fr(d_q + (size_t)(idx - 1) * 8, h);
        }
    }
}

bool launch_quotient_single(hipStream_t s, const uint32_t* d_coeffs, uint32_t n, const uint32_t z_mont[8], uint32_t* d_q,
                            uint32_t* d_small) {
    if (n == 0 || n > kPolySingleMax) return false;
    namespace hf = kzg_host;
    PolyPowers pw;
    std::memset(&pw, 0, sizeof pw);
    auto put = [](FrArg& dst, const hf::Fr& v) { std::memcpy(dst.l, v.l, 32); };
    hf::Fr z;
    std::memcpy(z.l, z_mont, 32);
    put(pw.z, z);
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
    auto put = [](FrArg& dst, const hf::Fr& v) { std::memcpy(dst.l, v.l, 32); };
    hf::Fr z;
    std::memcpy(z.l, z_mont, 32);
    put(pw.z, z);
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
    hipLaunchKernelGGL(k_poly_chunks, dim3(nblocks), dim3(kPolyBlock), 0, s, d_coeffs, n, pw, sc.d_chunk, sc.d_block, sc.d_flags);
    hipLaunchKernelGGL(k_poly_blocks, dim3(1), dim3(kPolyBlock), 0, s, sc.d_block, nblocks, pw, sc.d_result);
    if (d_q && n > 1)
        hipLaunchKernelGGL(k_poly_apply, dim3(nblocks), dim3(kPolyBlock), 0, s, d_coeffs, n, pw, sc.d_chunk, sc.d_block, d_q);
}

}  // namespace kzg
