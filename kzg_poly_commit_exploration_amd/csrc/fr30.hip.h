// fr30.hip.h -- BLS12-381 scalar field Fr in a SIGNED radix-2^30 representation (9 int32 digits), Montgomery
// products with R'' = 2^270, built for gfx950's v_mad_i64_i32 like field30.hip.h does for the base field.
//
// Why: an 8 x u32 Montgomery multiplier (rounds 1-2; tools/legacy_field/field.hip.h) is a carry chain -- per product 120 v_mad_u64_u32, 125 v_addc,
// ~140 moves and ~90 s_nop for the carry-out hazards, ~500 instructions -- and the quotient scan and the scalar recoding
// are bound by it.  With balanced 30-bit digits a column of the product (<= 9 digit products of a*b and 9 of m*r) stays
// inside a signed 64-bit accumulator:
//     9 * 2 * 2^58 (a*b, one operand may be a raw sum of two)  +  2^29 * sum|r_i| = 3.7 * 2^58 (m*r)  <  2^63,
// so a product is 81 + 81 plain multiply-adds and ~60 simple instructions.  r = 1 (mod 2^30), so the Montgomery digit
// is just the negated low digit of the column.
//
// What the forms mean.  The ABI keeps scalars as blst_fr images: x * 2^256 mod r in 8 x u32, fully reduced.  The device
// never changes that form: fr30_mul(a, b) = a * b / 2^270, so with ONE operand prepared on the host as w * 2^270 (every
// multiplier of the scans is a power of z that the host prepares, fr30_arg_from_mont256) the other operand and the
// result stay "times 2^256".  Loading is a re-slicing of the 256-bit integer into digits, storing is a carry pass and a
// conditional +-r (fr30_to_limbs).  Values are lazy integers: fr30_mul returns |v| <= 0.5001 r whatever the size of its
// operands (|a| |b| < 2^528), sums are digit-wise and carry-normalised (fr30_norm) before they feed a product.
//
// Replaces blst's Fr arithmetic behind Scalar::add / mul (reference src/scalar.rs:55-81) inside Polynomial::evaluate and
// divide_by_root (src/polynomial.rs:112-195), behind Scalar::to_le_bytes in front of every scalar multiplication
// (src/scalar.rs:83-93: msm_sort.hip's load_scalar) and behind the powers of the secret (src/trusted_setup.rs:50).
#pragma once
#include <stdint.h>

#ifndef KZG_HD
#ifdef __HIPCC__
#define KZG_HD __host__ __device__ __forceinline__
#else
#define KZG_HD inline
#endif
#endif

namespace kzg {

constexpr int kR9 = 9;           // digits
constexpr int kR9Bits = 30;      // radix 2^30
constexpr uint32_t kR9Mask = (1u << kR9Bits) - 1u;

struct Fr30 {
    int32_t d[kR9];
};

// r in balanced digits: r = sum RD[i] * 2^(30 i)  (tools/gen_field30_constants.py prints them)
KZG_HD constexpr int32_t fr30_rd(int i) {
    constexpr int32_t RD[9] = {0x1, -0x4, -0x1a4010, -0x1096ff40, -0x1e27faac, -0x189fdfd9, 0x17d48334, -0x162b3599, 0x73ee};
    return RD[i];
}
// r in unsigned digits (the canonical comparison of fr30_to_limbs)
KZG_HD constexpr uint32_t fr30_ru(int i) {
    constexpr uint32_t RU[9] = {0x1, 0x3ffffffc, 0x3fe5bfef, 0x2f6900bf, 0x21d80553, 0x27602026, 0x17d48333, 0x29d4ca67, 0x73ed};
    return RU[i];
}

// 2^270 mod r (the one of variable x variable products, where both operands carry the factor 2^270) and 2^540 mod r (takes a
// plain integer into that form), balanced digits of the centred residues
KZG_HD constexpr int32_t fr30_one270(int i) {
    constexpr int32_t C[9] = {-0x8d54, 0x23550, -0x21a2ac0, 0x1c22013a, -0x15d0deee, -0x1d3fc534, -0x1dfe7aaf, 0x12b2a695, 0x10dc};
    return C[i];
}
KZG_HD constexpr int32_t fr30_r2_540(int i) {
    constexpr int32_t C[9] = {0xefe9ec3, 0x1022c0f, 0x12313d61, 0x83bec60, -0xcc084be, 0x4a39dc, -0x17165989, 0x1dd02cf9, -0xc8f};
    return C[i];
}

KZG_HD int32_t fr30_sext30(uint32_t v) { return (int32_t)(v << 2) >> 2; }  // low 30 bits as a balanced digit

KZG_HD Fr30 fr30_zero() {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < kR9; i++) r.d[i] = 0;
    return r;
}
KZG_HD Fr30 fr30_add_raw(const Fr30& a, const Fr30& b) {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < kR9; i++) r.d[i] = a.d[i] + b.d[i];
    return r;
}
// one parallel carry pass: digits of magnitude < 2^31 - 2^29 come back into [-2^29 - 4, 2^29 + 4]; the value is unchanged
KZG_HD Fr30 fr30_norm(const Fr30& a) {
    Fr30 r;
    int32_t c[kR9 - 1];
#pragma unroll
    for (int i = 0; i < kR9 - 1; i++) c[i] = (a.d[i] + (1 << (kR9Bits - 1))) >> kR9Bits;
    r.d[0] = a.d[0] - (int32_t)((uint32_t)c[0] << kR9Bits);
#pragma unroll
    for (int i = 1; i < kR9 - 1; i++) r.d[i] = a.d[i] - (int32_t)((uint32_t)c[i] << kR9Bits) + c[i - 1];
    r.d[kR9 - 1] = a.d[kR9 - 1] + c[kR9 - 2];
    return r;
}
KZG_HD Fr30 fr30_add(const Fr30& a, const Fr30& b) { return fr30_norm(fr30_add_raw(a, b)); }

// a * b / 2^270 (mod r), product scanning.  Digits of both operands within [-2^29 - 4, 2^29 + 4] (one of them may be a
// raw sum of two such values); result digits 0..7 in [-2^29, 2^29), |result| <= 0.5001 r + |a b| / 2^270.
KZG_HD Fr30 fr30_mul(const Fr30& a, const Fr30& b) {
    int32_t m[kR9];
    Fr30 r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < kR9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (int64_t)a.d[i] * b.d[k - i];
#pragma unroll
        for (int j = 0; j < k; j++) acc += (int64_t)m[j] * fr30_rd(k - j);
        m[k] = fr30_sext30(0u - (uint32_t)acc);  // -r^-1 = -1 (mod 2^30)
        acc += (int64_t)m[k];                    // r_0 = 1: the low 30 bits are now zero
        acc >>= kR9Bits;
    }
#pragma unroll
    for (int k = kR9; k < 2 * kR9 - 1; k++) {
#pragma unroll
        for (int i = k - kR9 + 1; i < kR9; i++) acc += (int64_t)a.d[i] * b.d[k - i];
#pragma unroll
        for (int j = k - kR9 + 1; j < kR9; j++) acc += (int64_t)m[j] * fr30_rd(k - j);
        r.d[k - kR9] = fr30_sext30((uint32_t)acc);
        acc = (acc + (int64_t)(1 << (kR9Bits - 1))) >> kR9Bits;
    }
    r.d[kR9 - 1] = (int32_t)acc;
    return r;
}

// the 256-bit integer in 8 x u32 (any value below 2^256) as UNSIGNED digits (< 2^30): pure re-slicing.  Good enough as
// the operand of a product whose other operand is a constant of balanced digits (the columns stay below 2^63 / 2).
KZG_HD Fr30 fr30_from_limbs_raw(const uint32_t l[8]) {
    Fr30 u;
    u.d[0] = (int32_t)(l[0] & kR9Mask);
    u.d[1] = (int32_t)(((l[0] >> 30) | (l[1] << 2)) & kR9Mask);
    u.d[2] = (int32_t)(((l[1] >> 28) | (l[2] << 4)) & kR9Mask);
    u.d[3] = (int32_t)(((l[2] >> 26) | (l[3] << 6)) & kR9Mask);
    u.d[4] = (int32_t)(((l[3] >> 24) | (l[4] << 8)) & kR9Mask);
    u.d[5] = (int32_t)(((l[4] >> 22) | (l[5] << 10)) & kR9Mask);
    u.d[6] = (int32_t)(((l[5] >> 20) | (l[6] << 12)) & kR9Mask);
    u.d[7] = (int32_t)(((l[6] >> 18) | (l[7] << 14)) & kR9Mask);
    u.d[8] = (int32_t)(l[7] >> 16);
    return u;
}
// ... as balanced digits: one carry pass on top
KZG_HD Fr30 fr30_from_limbs(const uint32_t l[8]) { return fr30_norm(fr30_from_limbs_raw(l)); }

// |v| as a 256-bit integer in 8 x u32 and the sign of v, for a lazy value of magnitude below 2^256 (the scalar recoding:
// v is a product, |v| <= r / 2 + r / 2^31, and its sign moves onto the point).  Sequential carry to unsigned digits with a
// signed top digit, two's complement when negative.
KZG_HD bool fr30_abs_to_limbs(const Fr30& a, uint32_t l[8]) {
    uint32_t u[kR9];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < kR9 - 1; i++) {
        const int32_t t = a.d[i] + c;
        u[i] = (uint32_t)t & kR9Mask;
        c = t >> kR9Bits;  // floor
    }
    const int32_t top = a.d[kR9 - 1] + c;  // signed: the sign of the whole value
    const bool neg = top < 0;
    const uint32_t flip = neg ? kR9Mask : 0u;
    uint32_t carry = neg ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < kR9 - 1; i++) {
        const uint32_t t = (u[i] ^ flip) + carry;
        u[i] = t & kR9Mask;
        carry = t >> kR9Bits;
    }
    const uint32_t t8 = (neg ? ~(uint32_t)top : (uint32_t)top) + carry;
    l[0] = u[0] | (u[1] << 30);
    l[1] = (u[1] >> 2) | (u[2] << 28);
    l[2] = (u[2] >> 4) | (u[3] << 26);
    l[3] = (u[3] >> 6) | (u[4] << 24);
    l[4] = (u[4] >> 8) | (u[5] << 22);
    l[5] = (u[5] >> 10) | (u[6] << 20);
    l[6] = (u[6] >> 12) | (u[7] << 18);
    l[7] = (u[7] >> 14) | (t8 << 16);
    return neg;
}

// The canonical residue in [0, r) as 8 x u32, for a lazy value in (-r, 2r): what the scans hold where a result leaves
// them (a product, |v| <= 0.5001 r, plus at most one canonical coefficient).  Sequential carry to unsigned digits with a
// signed top digit, + r when negative, - r when that is not below r.
KZG_HD void fr30_to_limbs(const Fr30& a, uint32_t l[8]) {
    uint32_t u[kR9];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < kR9 - 1; i++) {
        const int32_t t = a.d[i] + c;
        u[i] = (uint32_t)t & kR9Mask;
        c = t >> kR9Bits;  // floor
    }
    int32_t top = a.d[kR9 - 1] + c;  // signed: the sign of the whole value
    // + r when negative
    const uint32_t neg = top < 0 ? 0xffffffffu : 0u;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < kR9 - 1; i++) {
        const uint32_t t = u[i] + (fr30_ru(i) & neg) + carry;
        u[i] = t & kR9Mask;
        carry = t >> kR9Bits;
    }
    top += (int32_t)((fr30_ru(kR9 - 1) & neg) + carry);
    // - r unless the value is below r already: compute v - r, keep it when it is not negative
    uint32_t w[kR9];
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < kR9 - 1; i++) {
        const int32_t t = (int32_t)u[i] - (int32_t)fr30_ru(i) + borrow;
        w[i] = (uint32_t)t & kR9Mask;
        borrow = t >> kR9Bits;  // 0 or -1
    }
    const int32_t wtop = top - (int32_t)fr30_ru(kR9 - 1) + borrow;
    const bool ge = wtop >= 0;
#pragma unroll
    for (int i = 0; i < kR9 - 1; i++) u[i] = ge ? w[i] : u[i];
    const uint32_t t8 = (uint32_t)(ge ? wtop : top);
    l[0] = u[0] | (u[1] << 30);
    l[1] = (u[1] >> 2) | (u[2] << 28);
    l[2] = (u[2] >> 4) | (u[3] << 26);
    l[3] = (u[3] >> 6) | (u[4] << 24);
    l[4] = (u[4] >> 8) | (u[5] << 22);
    l[5] = (u[5] >> 10) | (u[6] << 20);
    l[6] = (u[6] >> 12) | (u[7] << 18);
    l[7] = (u[7] >> 14) | (t8 << 16);
}

// the digits {c, 0, ..., 0} of a small constant: fr30_mul(x * 2^256, fr30_small(1 << 14)) = x, fr30_mul(x * 2^270, fr30_small(1)) = x
KZG_HD Fr30 fr30_small(int32_t c) {
    Fr30 r = fr30_zero();
    r.d[0] = c;
    return r;
}
KZG_HD Fr30 fr30_const_one270() {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < kR9; i++) r.d[i] = fr30_one270(i);
    return r;
}
KZG_HD Fr30 fr30_const_r2_540() {
    Fr30 r;
#pragma unroll
    for (int i = 0; i < kR9; i++) r.d[i] = fr30_r2_540(i);
    return r;
}

}  // namespace kzg
