// srs_io.hip -- SRS ingestion from the wire / on-disk forms, on the device:
//  * affine points (x, y as blst_fp, 96 bytes): what the binary cache of kzg_srs_save holds -- no normalisation;
//  * compressed points (48 bytes, ZCash encoding: reference src/curves.rs:99-183, the strings of the CLI's
//    setup.json): every lane decompresses one point, y = (x^3 + 4)^((p+1)/4) with the sign bit of the encoding,
//    instead of one blst_p1_uncompress per point on the host.
// Both write table level 0 in the table's own record form (x digits in words 0..12, y digits in words 16..28 of a 128-byte
// record, all zero = infinity); srs_kernels.hip builds the other levels from it.
#include "engine.h"
#include "field30.hip.h"

namespace kzg {

__device__ __forceinline__ void store_digits16(uint4* __restrict__ p, const Fq& a) {
    p[0] = make_uint4((uint32_t)a.d[0], (uint32_t)a.d[1], (uint32_t)a.d[2], (uint32_t)a.d[3]);
    p[1] = make_uint4((uint32_t)a.d[4], (uint32_t)a.d[5], (uint32_t)a.d[6], (uint32_t)a.d[7]);
    p[2] = make_uint4((uint32_t)a.d[8], (uint32_t)a.d[9], (uint32_t)a.d[10], (uint32_t)a.d[11]);
    p[3] = make_uint4((uint32_t)a.d[12], 0u, 0u, 0u);
}

// 96-byte affine records (x, y as blst_fp: 12 x u32, Montgomery R = 2^384) -> 128-byte table records; (0, 0) = infinity
__global__ void __launch_bounds__(256) k_affine96_to_table(const uint4* __restrict__ in, uint32_t n, uint4* __restrict__ table) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4* src = in + (size_t)i * 6;
    uint4* dst = table + (size_t)i * kAffineU4;
    uint32_t w[24];
#pragma unroll
    for (int t = 0; t < 6; t++) {
        const uint4 v = src[t];
        w[4 * t] = v.x; w[4 * t + 1] = v.y; w[4 * t + 2] = v.z; w[4 * t + 3] = v.w;
    }
    uint32_t any = 0;
#pragma unroll
    for (int t = 0; t < 24; t++) any |= w[t];
    if (!any) {
        const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 8; t++) dst[t] = zero;
        return;
    }
    // stored s = x * 2^384; the signed form wants x * 2^390 = 64 s: the same bits six places higher, then one product
    // with the Montgomery one to bring the magnitude below 0.62 p
    store_digits16(dst, fq_mul(fq_from_u32x12(w), fq_one()));
    store_digits16(dst + 4, fq_mul(fq_from_u32x12(w + 12), fq_one()));
}

KZG_HD Fq fq_const_2_780() {
    Fq c;
    constexpr int32_t V[13] = {
#include "field30_c780.inc"
    };
#pragma unroll
    for (int i = 0; i < kQ; i++) c.d[i] = V[i];
    return c;
}
KZG_HD Fq fq_const_half() {
    Fq c;
    constexpr int32_t V[13] = {
#include "field30_half.inc"
    };
#pragma unroll
    for (int i = 0; i < kQ; i++) c.d[i] = V[i];
    return c;
}
// plain integer behind a lazy Montgomery value, canonical balanced digits in [0, p)
__device__ __forceinline__ Fq fq_canonical_integer(const Fq& a) {
    Fq raw_one = fq_zero();
    raw_one.d[0] = 1;
    Fq t = fq_canon_digits(fq_mul(a, raw_one));  // x * 2^390 * 1 / 2^390 = x, |.| < 0.62 p
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < kQ; i++) s = t.d[i] != 0 ? t.d[i] : s;  // sign = sign of the most significant non-zero digit
    if (s < 0) {
#pragma unroll
        for (int i = 0; i < kQ; i++) t.d[i] += fq_pd(i);
        t = fq_canon_digits(t);
    }
    return t;
}
// a > b for canonical balanced digit vectors of non-negative integers
__device__ __forceinline__ bool fq_digits_greater(const Fq& a, const Fq& b) {
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < kQ; i++) {
        const int32_t d = a.d[i] - b.d[i];
        s = d != 0 ? d : s;
    }
    return s > 0;
}

// status[0]: index + 1 of the first malformed point (0 = all good), by atomicMin on index + 1 (pre-set to 0xffffffff)
__global__ void __launch_bounds__(64) k_uncompress(const uint8_t* __restrict__ in, uint32_t n, uint4* __restrict__ table,
                                                   uint32_t* __restrict__ status) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + (size_t)i * 48;
    uint4* dst = table + (size_t)i * kAffineU4;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    const uint32_t flags = b[0];
    const bool compressed = flags & 0x80, infinity = flags & 0x40, y_big = flags & 0x20;
    bool bad = !compressed;
    // big-endian 381-bit x -> twelve little-endian 32-bit words
    uint32_t w[12];
#pragma unroll
    for (int t = 0; t < 12; t++) {
        const uint8_t* q = b + 44 - 4 * t;
        uint32_t v = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
        if (t == 11) v &= 0x1fffffffu;
        w[t] = v;
    }
    uint32_t any = 0;
#pragma unroll
    for (int t = 0; t < 12; t++) any |= w[t];
    if (infinity) {
        bad = bad || any != 0 || y_big;
#pragma unroll
        for (int t = 0; t < 8; t++) dst[t] = zero;
        if (bad) atomicMin(status, i + 1);
        return;
    }
    // x < p ?
    {
        uint32_t borrow = 0;
#pragma unroll
        for (int t = 0; t < 12; t++) {
            constexpr uint32_t PW[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                         0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
            const uint64_t d = (uint64_t)w[t] - PW[t] - borrow;
            borrow = (uint32_t)(d >> 63);
        }
        bad = bad || borrow == 0;  // no borrow: x >= p
    }
    // plain integer x -> signed digits -> Montgomery form: x_digits * 2^780 / 2^390 = x * 2^390
    const Fq c = fq_const_2_780();
    Fq x;
    {
        Fq xi;
        uint32_t u[kQ];
#pragma unroll
        for (int k = 0; k < kQ; k++) {
            const int lo = 30 * k;
            const int wi = lo >> 5, sh = lo & 31;
            const uint64_t two = (uint64_t)(wi < 12 ? w[wi] : 0u) | ((uint64_t)(wi + 1 < 12 ? w[wi + 1] : 0u) << 32);
            u[k] = (uint32_t)(two >> sh) & (uint32_t)kQMask;
        }
        int32_t cy = 0;
#pragma unroll
        for (int k = 0; k < kQ - 1; k++) {
            const int32_t t = (int32_t)u[k] + cy;
            cy = (t + (1 << (kQBits - 1))) >> kQBits;
            xi.d[k] = t - (int32_t)((uint32_t)cy << kQBits);
        }
        xi.d[kQ - 1] = (int32_t)u[kQ - 1] + cy;
        x = fq_mul(xi, c);                           // x * 2^780 / 2^390 = x * 2^390
    }
    // t = x^3 + 4
    Fq four = fq_zero();
    four.d[0] = 4;
    four = fq_mul(four, c);  // 4 * 2^390, reduced
    const Fq t = fq_norm(fq_add_raw(fq_mul(fq_sqr(x), x), four));
    // y = t^((p + 1) / 4): p = 3 mod 4, so this is a square root whenever one exists
    constexpr uint64_t E[6] = {0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL,
                               0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL};
    Fq y = fq_one();
#pragma unroll 1
    for (int k = 378; k >= 0; k--) {
        y = fq_sqr(y);
        if ((E[k >> 6] >> (k & 63)) & 1) y = fq_mul(y, t);
    }
    bad = bad || !fq_is_zero(fq_norm(fq_sub_raw(fq_sqr(y), t)));  // not on the curve
    // the encoding's sign bit: set iff y > (p - 1) / 2 as an integer
    const Fq yi = fq_canonical_integer(y);
    const bool is_big = fq_digits_greater(yi, fq_const_half());
    if (is_big != y_big) y = fq_neg(y);
    store_digits16(dst, x);
    store_digits16(dst + 4, fq_norm(y));  // (the negation keeps the digits' size; one carry pass for the table's contract)
    if (bad) atomicMin(status, i + 1);
}

void launch_affine96_to_table(hipStream_t s, const void* d_affine96, uint32_t n, void* d_table) {
    if (!n) return;
    hipLaunchKernelGGL(k_affine96_to_table, dim3((n + 255) / 256), dim3(256), 0, s, (const uint4*)d_affine96, n, (uint4*)d_table);
}
void launch_uncompress(hipStream_t s, const void* d_compressed, uint32_t n, void* d_table, uint32_t* d_status) {
    if (!n) return;
    hipLaunchKernelGGL(k_uncompress, dim3((n + 63) / 64), dim3(64), 0, s, (const uint8_t*)d_compressed, n, (uint4*)d_table, d_status);
}

}  // namespace kzg
