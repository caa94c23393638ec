// srs_kernels.hip -- everything that happens once per SRS: ingest, window tables, trusted setup.  All of it in the
// accumulation kernel's own field (signed radix 2^30, field30.hip.h / g1_30.hip.h): the table is born in its final
// 128-byte record form (x digits in words 0..12, y digits in words 16..28, all zero = infinity).
//
//  * ingest: the reference's SRS entries are blst_p1 Jacobian points with arbitrary Z
//    (src/trusted_setup.rs:54-62); the accumulation kernel wants affine points (mixed addition),
//    so they are normalised here with a per-lane batched inversion (Montgomery's trick, one safegcd
//    inversion per 32 points).
//  * window tables: T[j][i] = 2^(c j) SRS[i].  With them every window of the signed-digit recoding
//    lands in ONE set of 2^(c-1) buckets, so a commitment pays the bucket reduction once instead of
//    once per window and needs no window-combining doublings.  HBM is 288 GB; the table for 2^20
//    points and 15 windows is 1.9 GiB.
//  * trusted setup on the device (reference src/trusted_setup.rs:40-62, G1 side): SRS[i] = [s^i]G1 by
//    fixed-base windowing over a table of d * 2^(8w) * G1.
#include "engine.h"
#include "fr30.hip.h"  // Fr: the powers of the secret
#include "field30_inv.hip.h"
#include "g1_30.hip.h"

namespace kzg {

#ifndef KZG_DEV
#define KZG_DEV __device__ __forceinline__
#endif

constexpr int kNormK = 32;  // points per lane in the batched inversion

// blst_fp words (12 x u32, Montgomery R = 2^384, canonical) -> the signed field's Montgomery form, reduced below 0.62 p
KZG_DEV Fq fq_from_blst_words(const uint4* __restrict__ p) {
    const uint4 a = p[0], b = p[1], c = p[2];
    const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    return fq_mul(fq_from_u32x12(w), fq_one());
}
KZG_DEV bool words_all_zero(const uint4* __restrict__ p, int n4) {
    uint32_t any = 0;
    for (int i = 0; i < n4; i++) any |= p[i].x | p[i].y | p[i].z | p[i].w;
    return any == 0;
}
KZG_DEV void store_table_point(uint4* __restrict__ rec, const Fq& x, const Fq& y) {
    store_fq16(rec, x);
    store_fq16(rec + 4, y);
}
KZG_DEV void store_table_infinity(uint4* __restrict__ rec) {
    const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 8; t++) rec[t] = zero;
}
KZG_DEV Affine30 load_table_point(const uint4* __restrict__ rec) {
    Affine30 a;
    a.x = load_fq16(rec);
    a.y = load_fq16(rec + 4);
    return a;
}
// the inversion is called once per lane: a real call keeps its ~100 registers of state out of the loops around it
static __device__ __noinline__ void fq_inv_inplace(Fq* io) {
    const Fq a = *io;
    *io = fq_inv(a);
}

// MODE 0: records are blst_p1 Jacobian {X, Y, Z} (144 B, blst's words): x = X / Z^2, y = Y / Z^3
// MODE 1: records are XYZZ30 {X, Y, ZZ, ZZZ} (256 B, digits):           x = X / ZZ,  y = Y / ZZZ
// prefix: one 64-byte digit record per point (the running product of the denominators)
template <int MODE>
__global__ void __launch_bounds__(64) k_normalize(const uint4* __restrict__ in, uint32_t n, uint4* __restrict__ out,
                                                  uint4* __restrict__ prefix) {
    constexpr int REC = MODE == 0 ? 9 : 16;  // uint4 per record
    constexpr int DEN = MODE == 0 ? 6 : 12;  // uint4 offset of the denominator (Z or ZZZ)
    constexpr int DEN4 = MODE == 0 ? 3 : 4;  // uint4 of the denominator
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = t * kNormK;
    if (lo >= n) return;
    const uint32_t hi = lo + kNormK < n ? lo + kNormK : n;
    auto denominator = [&](const uint4* rec) { return MODE == 0 ? fq_from_blst_words(rec + DEN) : load_fq16(rec + DEN); };
    Fq run = fq_one();
    for (uint32_t i = lo; i < hi; i++) {
        const uint4* rec = in + (size_t)i * REC;
        if (!words_all_zero(rec + DEN, DEN4)) run = fq_mul(run, denominator(rec));  // (infinity is only ever written as exact zeros)
        store_fq16(prefix + (size_t)i * 4, run);
    }
    Fq inv = run;
    fq_inv_inplace(&inv);
    for (uint32_t i = hi; i-- > lo;) {
        const uint4* rec = in + (size_t)i * REC;
        uint4* o = out + (size_t)i * kAffineU4;
        if (words_all_zero(rec + DEN, DEN4)) {  // infinity -> all zero
            store_table_infinity(o);
            continue;
        }
        const Fq d = denominator(rec);
        const Fq prev = i == lo ? fq_one() : load_fq16(prefix + (size_t)(i - 1) * 4);
        const Fq di = fq_mul(inv, prev);  // 1 / d
        inv = fq_mul(inv, d);
        Fq x, y;
        if (MODE == 0) {
            const Fq di2 = fq_sqr(di);
            x = fq_mul(fq_from_blst_words(rec), di2);
            y = fq_mul(fq_from_blst_words(rec + 3), fq_mul(di2, di));
        } else {
            const Fq iz = fq_mul(load_fq16(rec + 8), di);  // ZZ / ZZZ = 1 / z
            x = fq_mul(load_fq16(rec), fq_sqr(iz));
            y = fq_mul(load_fq16(rec + 4), di);
        }
        store_table_point(o, x, y);
    }
}

KZG_DEV XYZZ30 xyzz30_from_affine(const Affine30& p) {
    XYZZ30 a = xyzz30_inf();
    if (fq_all_zero(p.x) && fq_all_zero(p.y)) return a;
    a.X = p.x;
    a.Y = p.y;
    a.ZZ = fq_one();
    a.ZZZ = a.ZZ;
    return a;
}

__global__ void __launch_bounds__(64) k_window_double(const uint4* __restrict__ prev, uint32_t n, uint32_t c,
                                                      uint4* __restrict__ out_xyzz) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ30 a = xyzz30_from_affine(load_table_point(prev + (size_t)i * kAffineU4));
#pragma unroll 1
    for (uint32_t k = 0; k < c; k++) xyzz30_dbl_body(a);
    store_xyzz30(out_xyzz + (size_t)i * kXyzzU4, a);
}

// ---- fixed-base trusted setup -----------------------------------------------------------------
constexpr int kGWindows = 32;   // 8-bit windows over a 256-bit scalar
constexpr int kGDigits = 255;   // digits 1..255
size_t srs_gtable_bytes() { return (size_t)kGWindows * kGDigits * kAffineBytes; }

KZG_DEV Affine30 g1_generator30() {  // blst's Montgomery words of the generator, converted
    constexpr uint32_t GX[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                                 0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
    constexpr uint32_t GY[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                                 0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
    uint32_t gx[12], gy[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        gx[i] = GX[i];
        gy[i] = GY[i];
    }
    Affine30 g;
    g.x = fq_mul(fq_from_u32x12(gx), fq_one());
    g.y = fq_mul(fq_from_u32x12(gy), fq_one());
    return g;
}

// entry (w, d-1) = d * 2^(8w) * G, as XYZZ (normalised afterwards)
__global__ void __launch_bounds__(64) k_gtable_build(uint4* __restrict__ out_xyzz) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint32_t)(kGWindows * kGDigits)) return;
    const uint32_t w = e / kGDigits, d = e % kGDigits + 1;
    const Affine30 g = g1_generator30();
    XYZZ30 acc = xyzz30_inf();
#pragma unroll 1
    for (int b = 7; b >= 0; b--) {
        xyzz30_dbl_body(acc);
        if ((d >> b) & 1) xyzz30_madd(acc, g, false);
    }
#pragma unroll 1
    for (uint32_t k = 0; k < 8 * w; k++) xyzz30_dbl_body(acc);
    store_xyzz30(out_xyzz + (size_t)e * kXyzzU4, acc);
}

struct FrArg8 {
    uint32_t l[8];
};

__global__ void __launch_bounds__(64) k_srs_points(FrArg8 secret, uint64_t first, uint32_t n,
                                                   const uint4* __restrict__ gtable, uint4* __restrict__ out_xyzz) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the secret arrives as the raw 256-bit integer: times 2^540 over the multiplier's 2^270 it is s * 2^270, reduced
    // (fr30.hip.h; with that factor on both operands a product keeps it)
    const Fr30 s = fr30_mul(fr30_from_limbs(secret.l), fr30_const_r2_540());
    // s^(first + i) by square-and-multiply (reference src/trusted_setup.rs:50 walks it sequentially)
    uint64_t e = first + i;
    Fr30 pw = fr30_const_one270();
    Fr30 base = s;
    while (e) {
        if (e & 1) pw = fr30_mul(pw, base);
        base = fr30_mul(base, base);
        e >>= 1;
    }
    struct {
        uint32_t l[8];
    } k;
    fr30_to_limbs(fr30_mul(pw, fr30_small(1)), k.l);  // canonical scalar, as Scalar::to_le_bytes (src/scalar.rs:83-93)
    XYZZ30 acc = xyzz30_inf();
#pragma unroll 1
    for (int w = 0; w < kGWindows; w++) {
        const uint32_t d = k.l[0] & 0xffu;
#pragma unroll
        for (int t = 0; t < 7; t++) k.l[t] = (k.l[t] >> 8) | (k.l[t + 1] << 24);
        k.l[7] >>= 8;
        if (d) xyzz30_madd(acc, load_table_point(gtable + (size_t)(w * kGDigits + (d - 1)) * kAffineU4), false);
    }
    store_xyzz30(out_xyzz + (size_t)i * kXyzzU4, acc);
}

void launch_jacobian_to_affine(hipStream_t s, const void* d_jac, uint32_t n, void* d_out, void* d_prefix) {
    if (!n) return;
    const uint32_t lanes = (n + kNormK - 1) / kNormK;
    hipLaunchKernelGGL(k_normalize<0>, dim3((lanes + 63) / 64), dim3(64), 0, s, (const uint4*)d_jac, n, (uint4*)d_out,
                       (uint4*)d_prefix);
}
static void launch_xyzz_to_affine(hipStream_t s, const void* d_xyzz, uint32_t n, void* d_out, void* d_prefix) {
    if (!n) return;
    const uint32_t lanes = (n + kNormK - 1) / kNormK;
    hipLaunchKernelGGL(k_normalize<1>, dim3((lanes + 63) / 64), dim3(64), 0, s, (const uint4*)d_xyzz, n, (uint4*)d_out,
                       (uint4*)d_prefix);
}
void launch_table_window(hipStream_t s, const void* d_prev, uint32_t n, uint32_t c, void* d_xyzz_tmp, void* d_prefix,
                         void* d_next) {
    if (!n) return;
    hipLaunchKernelGGL(k_window_double, dim3((n + 63) / 64), dim3(64), 0, s, (const uint4*)d_prev, n, c,
                       (uint4*)d_xyzz_tmp);
    launch_xyzz_to_affine(s, d_xyzz_tmp, n, d_next, d_prefix);
}
void launch_srs_generate(hipStream_t s, const uint32_t* secret_raw8, uint64_t first, uint32_t n, void* d_gtable,
                         void* d_xyzz_tmp, void* d_prefix, void* d_out) {
    if (!n) return;
    const uint32_t ge = kGWindows * kGDigits;
    // the temporaries are sized for max(n, ge) records by the caller
    hipLaunchKernelGGL(k_gtable_build, dim3((ge + 63) / 64), dim3(64), 0, s, (uint4*)d_xyzz_tmp);
    launch_xyzz_to_affine(s, d_xyzz_tmp, ge, d_gtable, d_prefix);
    FrArg8 sec;
    for (int i = 0; i < 8; i++) sec.l[i] = secret_raw8[i];
    hipLaunchKernelGGL(k_srs_points, dim3((n + 63) / 64), dim3(64), 0, s, sec, first, n, (const uint4*)d_gtable,
                       (uint4*)d_xyzz_tmp);
    launch_xyzz_to_affine(s, d_xyzz_tmp, n, d_out, d_prefix);
}
}  // namespace kzg
