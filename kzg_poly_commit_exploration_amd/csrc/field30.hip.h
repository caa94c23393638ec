// field30.hip.h -- BLS12-381 base field Fp in a SIGNED radix-2^30 representation (13 int32 digits),
// Montgomery form with R' = 2^390, built for gfx950's v_mad_i64_i32.
//
// Why: v_mad_u64_u32 has no carry-in, so with full 32-bit limbs (field_fips.hip.h) every multiply-add needs a
// v_addc to catch the carry out of its 64-bit accumulator: 288 + 288 instructions per product plus the moves.
// With digits d_i in [-2^29, 2^29] a product is below 2^58 in magnitude and a whole column of a Montgomery
// product (<= 13 digit products of a*b and 13 of m*p) stays inside a signed 64-bit accumulator:
//     12 * 2^59 (a*b, one operand may be an unreduced sum of two)  +  2^29 * sum|p_i| = 1.55 * 2^60 (m*p)  <  2^63,
// so a column is a plain chain of v_mad_i64_i32 on one register pair -- 338 multiply-adds and ~115 simple
// instructions per product, no carry bookkeeping at all.  Values are "lazy" integers v with |v| < 2^385
// congruent to the element times 2^390: additions and subtractions are digit-wise (no carries, no
// conditional subtraction of p), followed by one parallel carry pass before the digits feed a product.
//
// Contract of the digit sizes (checked exhaustively by tests/test_field30.py on the host build):
//   * fq_mul / fq_sqr outputs: digits 0..11 in [-2^29, 2^29), digit 12 small (|v| < 0.62 p).
//   * fq_mul inputs: max|a_i| * max|b_j| <= 2.1 * 2^58 over digits 0..11 (e.g. a weakly normalised, b a raw
//     sum of two weakly normalised values), |a|,|b| < 2^385.
//   * fq_norm (one parallel carry pass) brings digits of magnitude < 2^31 into [-2^29 - 4, 2^29 + 4].
//
// Replaces blst's Fp arithmetic behind the reference's G1Point::add / mult (src/curves.rs:79-96).
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

constexpr int kQ = 13;          // digits
constexpr int kQBits = 30;      // radix 2^30
constexpr int32_t kQMask = (1 << kQBits) - 1;

struct Fq {
    int32_t d[kQ];
};

// p in balanced digits: p = sum PD[i] * 2^(30 i), |PD[i]| < 2^29
KZG_HD constexpr int32_t fq_pd(int i) {
    constexpr int32_t PD[13] = {-0x5555,     -0x18040000, 0x153ffffc,  -0x15000054, -0xf09dbe1, 0x34a83db, 0x112bf673,
                                0x12e13ce1,  -0x13289b89, 0x1ed90d2f,  -0x165b4e46, -0x571a006, 0x1a0112};
    return PD[i];
}
constexpr uint32_t kQN0 = 0x3ffcfffdu;  // -p^-1 mod 2^30

KZG_HD int32_t fq_sext30(uint32_t v) { return (int32_t)(v << 2) >> 2; }  // low 30 bits as a balanced digit

// (acc - balanced low digit) >> 30  ==  (acc + 2^29) >> 30: one 64-bit add and one 64-bit shift
KZG_HD int64_t fq_round_shift(int64_t acc) { return (acc + (int64_t)(1 << (kQBits - 1))) >> kQBits; }

// KZG_F30_SERIAL_COLUMNS: keep the columns of a product in program order (one live accumulator) instead of letting
// the compiler run a dozen of them side by side -- fewer VGPRs, less instruction-level parallelism
#if defined(KZG_F30_SERIAL_COLUMNS) && defined(__HIP_DEVICE_COMPILE__)
#define KZG_F30_FENCE(acc) asm volatile("" : "+v"(acc))
#else
#define KZG_F30_FENCE(acc) ((void)0)
#endif

KZG_HD Fq fq_zero() {
    Fq r;
#pragma unroll
    for (int i = 0; i < kQ; i++) r.d[i] = 0;
    return r;
}

// digit-wise sum / difference / negation: no carries (digits grow by one bit)
KZG_HD Fq fq_add_raw(const Fq& a, const Fq& b) {
    Fq r;
#pragma unroll
    for (int i = 0; i < kQ; i++) r.d[i] = a.d[i] + b.d[i];
    return r;
}
KZG_HD Fq fq_sub_raw(const Fq& a, const Fq& b) {
    Fq r;
#pragma unroll
    for (int i = 0; i < kQ; i++) r.d[i] = a.d[i] - b.d[i];
    return r;
}
KZG_HD Fq fq_neg(const Fq& a) {
    Fq r;
#pragma unroll
    for (int i = 0; i < kQ; i++) r.d[i] = -a.d[i];
    return r;
}
KZG_HD Fq fq_cneg(const Fq& a, bool neg) {  // neg ? -a : a
    const int32_t s = neg ? -1 : 0;
    Fq r;
#pragma unroll
    for (int i = 0; i < kQ; i++) r.d[i] = (a.d[i] ^ s) - s;
    return r;
}

// one parallel carry pass: digits of magnitude < 2^31 - 2^29 come back into [-2^29 - 4, 2^29 + 4]
// (each carry is in [-4, 4]); the value is unchanged.  Digit 12 absorbs the last carry.
KZG_HD Fq fq_norm(const Fq& a) {
    Fq r;
    int32_t c[kQ - 1];
#pragma unroll
    for (int i = 0; i < kQ - 1; i++) c[i] = (a.d[i] + (1 << (kQBits - 1))) >> kQBits;
    r.d[0] = a.d[0] - (int32_t)((uint32_t)c[0] << kQBits);
#pragma unroll
    for (int i = 1; i < kQ - 1; i++) r.d[i] = a.d[i] - (int32_t)((uint32_t)c[i] << kQBits) + c[i - 1];
    r.d[kQ - 1] = a.d[kQ - 1] + c[kQ - 2];
    return r;
}
// the same for digits of any int32 magnitude (sums of four): the rounding carry is taken from d >> 1 so that
// adding the half does not overflow; identical result where both apply
KZG_HD Fq fq_norm_wide(const Fq& a) {
    Fq r;
    int32_t c[kQ - 1];
#pragma unroll
    for (int i = 0; i < kQ - 1; i++) c[i] = ((a.d[i] >> 1) + (1 << (kQBits - 2))) >> (kQBits - 1);
    r.d[0] = a.d[0] - (int32_t)((uint32_t)c[0] << kQBits);
#pragma unroll
    for (int i = 1; i < kQ - 1; i++) r.d[i] = a.d[i] - (int32_t)((uint32_t)c[i] << kQBits) + c[i - 1];
    r.d[kQ - 1] = a.d[kQ - 1] + c[kQ - 2];
    return r;
}
KZG_HD Fq fq_add(const Fq& a, const Fq& b) { return fq_norm(fq_add_raw(a, b)); }
KZG_HD Fq fq_sub(const Fq& a, const Fq& b) { return fq_norm(fq_sub_raw(a, b)); }

// full (sequential) carry pass: digits 0..11 in [-2^29, 2^29) exactly -- the unique balanced form of the integer
KZG_HD Fq fq_canon_digits(const Fq& a) {
    Fq r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < kQ - 1; i++) {
        int32_t t = a.d[i] + c;
        int32_t q = (t + (1 << (kQBits - 1))) >> kQBits;
        r.d[i] = t - (int32_t)((uint32_t)q << kQBits);
        c = q;
    }
    r.d[kQ - 1] = a.d[kQ - 1] + c;
    return r;
}

// Montgomery product a * b / 2^390 (mod p), product scanning; see the header for the digit contract.
KZG_HD Fq fq_mul(const Fq& a, const Fq& b) {
    int32_t m[kQ];
    Fq r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < kQ; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (int64_t)a.d[i] * b.d[k - i];
#pragma unroll
        for (int j = 0; j < k; j++) acc += (int64_t)m[j] * fq_pd(k - j);
        m[k] = fq_sext30((uint32_t)acc * kQN0);
        acc += (int64_t)m[k] * fq_pd(0);  // low 30 bits are now zero
        acc >>= kQBits;
        KZG_F30_FENCE(acc);
    }
#pragma unroll
    for (int k = kQ; k < 2 * kQ - 1; k++) {
#pragma unroll
        for (int i = k - kQ + 1; i < kQ; i++) acc += (int64_t)a.d[i] * b.d[k - i];
#pragma unroll
        for (int j = k - kQ + 1; j < kQ; j++) acc += (int64_t)m[j] * fq_pd(k - j);
        r.d[k - kQ] = fq_sext30((uint32_t)acc);
        acc = fq_round_shift(acc);
        KZG_F30_FENCE(acc);
    }
    r.d[kQ - 1] = (int32_t)acc;
    return r;
}

// (a * b - c * d) / 2^390 (mod p) with ONE Montgomery reduction: both digit products are accumulated into the same
// columns.  Column bound: 2 x 12 full products of at most (2^29 + 4)^2 plus the m * p part,
// 6.0 * 2^60 + 1.55 * 2^60 < 2^63 -- so ALL FOUR operands must be weakly normalised (|digit| <= 2^29 + 4; no raw sums
// here), |a b - c d| < 2^771.  Saves 169 multiply-adds and the carry work of one product wherever the group law
// subtracts two products (Y3 = R (Q - X3) - Y1 PPP).
KZG_HD Fq fq_mul_sub(const Fq& a, const Fq& b, const Fq& c, const Fq& d) {
    int32_t m[kQ];
    Fq r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < kQ; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (int64_t)a.d[i] * b.d[k - i];
#pragma unroll
        for (int i = 0; i <= k; i++) acc -= (int64_t)c.d[i] * d.d[k - i];
#pragma unroll
        for (int j = 0; j < k; j++) acc += (int64_t)m[j] * fq_pd(k - j);
        m[k] = fq_sext30((uint32_t)acc * kQN0);
        acc += (int64_t)m[k] * fq_pd(0);
        acc >>= kQBits;
        KZG_F30_FENCE(acc);
    }
#pragma unroll
    for (int k = kQ; k < 2 * kQ - 1; k++) {
#pragma unroll
        for (int i = k - kQ + 1; i < kQ; i++) acc += (int64_t)a.d[i] * b.d[k - i];
#pragma unroll
        for (int i = k - kQ + 1; i < kQ; i++) acc -= (int64_t)c.d[i] * d.d[k - i];
#pragma unroll
        for (int j = k - kQ + 1; j < kQ; j++) acc += (int64_t)m[j] * fq_pd(k - j);
        r.d[k - kQ] = fq_sext30((uint32_t)acc);
        acc = fq_round_shift(acc);
        KZG_F30_FENCE(acc);
    }
    r.d[kQ - 1] = (int32_t)acc;
    return r;
}

// Montgomery square: cross products once against the doubled operand (91 instead of 169 digit products)
KZG_HD Fq fq_sqr(const Fq& a) {
    int32_t m[kQ], dbl[kQ];
#pragma unroll
    for (int i = 0; i < kQ; i++) dbl[i] = a.d[i] * 2;
    Fq r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < kQ; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (int64_t)dbl[i] * a.d[k - i];
        if ((k & 1) == 0) acc += (int64_t)a.d[k / 2] * a.d[k / 2];
#pragma unroll
        for (int j = 0; j < k; j++) acc += (int64_t)m[j] * fq_pd(k - j);
        m[k] = fq_sext30((uint32_t)acc * kQN0);
        acc += (int64_t)m[k] * fq_pd(0);
        acc >>= kQBits;
        KZG_F30_FENCE(acc);
    }
#pragma unroll
    for (int k = kQ; k < 2 * kQ - 1; k++) {
#pragma unroll
        for (int i = k - kQ + 1; 2 * i < k; i++) acc += (int64_t)dbl[i] * a.d[k - i];
        if ((k & 1) == 0) acc += (int64_t)a.d[k / 2] * a.d[k / 2];
#pragma unroll
        for (int j = k - kQ + 1; j < kQ; j++) acc += (int64_t)m[j] * fq_pd(k - j);
        r.d[k - kQ] = fq_sext30((uint32_t)acc);
        acc = fq_round_shift(acc);
        KZG_F30_FENCE(acc);
    }
    r.d[kQ - 1] = (int32_t)acc;
    return r;
}

// v == 0 (mod p) for |v| < 3.5 p.  The integer is k*p with |k| <= 3; its residue mod 2^30 only depends on digit 0,
// so seven compares on one word reject everything but ~2^-27 of the non-zero values; the exact test runs then.
// (by value: a reference parameter of a real call would pin every tested element in private memory)
#if defined(__HIPCC__)
static __host__ __device__ __noinline__ bool fq_is_zero_slow(Fq a) {
#else
inline bool fq_is_zero_slow(Fq a) {
#endif
    const Fq c = fq_canon_digits(a);
    // k*p in canonical balanced digits, k = -3..3
    for (int k = -3; k <= 3; k++) {
        Fq kp;
#pragma unroll
        for (int i = 0; i < kQ; i++) kp.d[i] = k * fq_pd(i);
        kp = fq_canon_digits(kp);
        bool eq = true;
#pragma unroll
        for (int i = 0; i < kQ; i++) eq = eq && (kp.d[i] == c.d[i]);
        if (eq) return true;
    }
    return false;
}
KZG_HD bool fq_is_zero(const Fq& a) {
    const uint32_t lo = (uint32_t)a.d[0] & (uint32_t)kQMask;
    const uint32_t p0 = (uint32_t)fq_pd(0) & (uint32_t)kQMask;
    bool maybe = lo == 0;
#pragma unroll
    for (uint32_t k = 1; k <= 3; k++) {
        maybe = maybe || lo == ((k * p0) & (uint32_t)kQMask) || lo == ((0u - k * p0) & (uint32_t)kQMask);
    }
    if (!maybe) return false;
    return fq_is_zero_slow(a);
}

// ---- conversions with the library's storage format (12 x u32, Montgomery R = 2^384, value in [0, 2p)) ----------
// stored s = x * 2^384; the signed form wants x * 2^390 = 64 s, i.e. the same bits read six places higher.
// 64 * 2p < 2^388: inside the range of the representation (digit 12 < 2^28).  No reduction, no multiplication.
KZG_HD Fq fq_from_u32x12(const uint32_t* s) {
    // u[i] = bits [30 i - 6, 30 i + 24) of s (zero below bit 0): unsigned 30-bit digits of 64 s
    Fq r;
    uint32_t u[kQ];
#pragma unroll
    for (int i = 0; i < kQ; i++) {
        const int lo = 30 * i - 6;  // first bit of s in this digit
        if (lo < 0) {
            u[i] = (s[0] << 6) & (uint32_t)kQMask;
        } else {
            const int w = lo >> 5, sh = lo & 31;
            uint64_t two = (uint64_t)s[w] | ((uint64_t)(w + 1 < 12 ? s[w + 1] : 0u) << 32);
            u[i] = (uint32_t)(two >> sh) & (uint32_t)kQMask;
        }
    }
    // balance: digits >= 2^29 borrow from the next one
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < kQ - 1; i++) {
        int32_t t = (int32_t)u[i] + c;
        c = (t + (1 << (kQBits - 1))) >> kQBits;
        r.d[i] = t - (int32_t)((uint32_t)c << kQBits);
    }
    r.d[kQ - 1] = (int32_t)u[kQ - 1] + c;
    return r;
}

KZG_HD Fq fq_one() {  // 2^390 mod p, balanced digits (tools/gen_field30_constants.py)
    Fq c;
    constexpr int32_t V[13] = {
#include "field30_one.inc"
    };
#pragma unroll
    for (int i = 0; i < kQ; i++) c.d[i] = V[i];
    return c;
}

KZG_HD Fq fq_const_2_1170() {  // 2^1170 mod p: v^-1 (as an integer) times this, as a Montgomery product, is the inverse's Montgomery form
    Fq c;
    constexpr int32_t V[13] = {
#include "field30_c1170.inc"
    };
#pragma unroll
    for (int i = 0; i < kQ; i++) c.d[i] = V[i];
    return c;
}

// Montgomery R' -> R: multiply by 2^384 (as a Montgomery factor: the constant 2^384 mod p in digits), then make
// the value canonical in [0, p) and repack into twelve 32-bit words.
KZG_HD Fq fq_const_2_384() {
    // 2^384 mod p in balanced digits (generated by tools/gen_field30_constants.py)
    Fq c;
    constexpr int32_t V[13] = {
#include "field30_c384.inc"
    };
#pragma unroll
    for (int i = 0; i < kQ; i++) c.d[i] = V[i];
    return c;
}
KZG_HD void fq_to_u32x12(const Fq& a, uint32_t* out) {
    Fq t = fq_mul(a, fq_const_2_384());  // x * 2^384 as an integer in (-0.62 p, 0.62 p)
    // add p when negative, then emit unsigned digits
    t = fq_canon_digits(t);
    const bool negative = t.d[kQ - 1] < 0 || (t.d[kQ - 1] == 0 && [&] {
        // sign of a balanced number = sign of its most significant non-zero digit
        int32_t s = 0;
#pragma unroll
        for (int i = 0; i < kQ - 1; i++) s = t.d[i] != 0 ? t.d[i] : s;
        return s < 0;
    }());
    if (negative) {
#pragma unroll
        for (int i = 0; i < kQ; i++) t.d[i] += fq_pd(i);
    }
    // unsigned 30-bit digits
    uint32_t u[kQ];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < kQ - 1; i++) {
        int32_t v = t.d[i] + c;
        c = v >> kQBits;  // floor
        u[i] = (uint32_t)v & (uint32_t)kQMask;
    }
    u[kQ - 1] = (uint32_t)(t.d[kQ - 1] + c);
#pragma unroll
    for (int w = 0; w < 12; w++) {
        // bits [32 w, 32 w + 32) of sum u[i] 2^(30 i)
        const int lo = 32 * w;
        const int i0 = lo / 30, sh = lo - 30 * i0;
        uint64_t two = (uint64_t)u[i0] | ((uint64_t)(i0 + 1 < kQ ? u[i0 + 1] : 0u) << 30) |
                       ((uint64_t)(i0 + 2 < kQ ? u[i0 + 2] : 0u) << 60);
        out[w] = (uint32_t)(two >> sh);
    }
}

}  // namespace kzg
