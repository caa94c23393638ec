// host_field.hpp -- host-side Fp / G1 arithmetic of the PRODUCT library (not the oracle).
//
// The device reduces an MSM to a few dozen partial sums; the host finishes the job the way the
// reference's caller would with blst on the CPU: a handful of complete point additions
// (blst_p1_add_or_double, reference src/curves.rs:79-85), one inversion to normalise, and the
// ZCash compression the reference's serde emits (src/curves.rs:99-110).  A single GPU lane needs
// ~1 us per 381-bit multiplication, a host core ~40 ns, so this serial tail belongs here.
// Also used for the multi-GPU combine of the RCCL-gathered partial sums.
#pragma once
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <vector>

#include "field30_inv.hip.h"

namespace kzg_host {

typedef unsigned __int128 u128;

struct Fp {
    uint64_t l[6];
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3] | l[4] | l[5]) == 0; }
    bool operator==(const Fp& o) const { return std::memcmp(l, o.l, sizeof l) == 0; }
};

static const Fp kP = {{0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                       0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL}};
static const Fp kOne = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                         0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}};
static const uint64_t kN0 = 0x89f3fffcfffcfffdULL;
static const Fp kHalf = {{0xdcff7fffffffd555ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL,
                          0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL}};

inline bool geq(const Fp& a, const Fp& b) {
    for (int i = 5; i >= 0; --i)
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    return true;
}
inline Fp raw_sub(const Fp& a, const Fp& b, uint64_t& borrow) {
    Fp r;
    borrow = 0;
    for (int i = 0; i < 6; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return r;
}
inline Fp raw_add(const Fp& a, const Fp& b) {
    Fp r;
    u128 c = 0;
    for (int i = 0; i < 6; ++i) {
        c += (u128)a.l[i] + b.l[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
    }
    return r;
}
inline Fp operator+(const Fp& a, const Fp& b) {
    Fp s = raw_add(a, b);
    uint64_t br;
    return geq(s, kP) ? raw_sub(s, kP, br) : s;
}
inline Fp operator-(const Fp& a, const Fp& b) {
    uint64_t br;
    Fp d = raw_sub(a, b, br);
    return br ? raw_add(d, kP) : d;
}
inline Fp neg(const Fp& a) {
    uint64_t br;
    return a.is_zero() ? a : raw_sub(kP, a, br);
}
// Montgomery product a * b / 2^384 mod p, coarsely integrated operand scanning (CIOS), fully unrolled by the
// compiler.  Accepts any a < 2^384 and b < p (the result of an unreduced addition may enter as a): used where an
// operand is not reduced (fp_from_digits30); operator* below is the product of two reduced elements.
inline Fp mul_wide(const Fp& a, const Fp& b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; ++i) {
        u128 c = 0;
        for (int j = 0; j < 6; ++j) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[6] = (uint64_t)c;
        t[7] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * kN0;
        c = ((u128)m * kP.l[0] + t[0]) >> 64;
        for (int j = 1; j < 6; ++j) {
            c += (u128)m * kP.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[5] = (uint64_t)c;
        t[6] = t[7] + (uint64_t)(c >> 64);
    }
    Fp r = {{t[0], t[1], t[2], t[3], t[4], t[5]}};
    uint64_t br;
    if (t[6] || geq(r, kP)) r = raw_sub(r, kP, br);
    return r;
}
// a, b < p.  p < 2^383 leaves the top bit of every partial sum free, so the two carry words of the general routine
// disappear (the "no-carry" CIOS): each round is two interleaved multiply-accumulate chains over six limbs.
inline Fp mul_portable(const Fp& a, const Fp& b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    for (int i = 0; i < 6; ++i) {
        const uint64_t bi = b.l[i];
        u128 A = (u128)a.l[0] * bi + t0;
        const uint64_t m = (uint64_t)A * kN0;
        u128 C = (u128)m * kP.l[0] + (uint64_t)A;
        A = (u128)a.l[1] * bi + t1 + (uint64_t)(A >> 64);
        C = (u128)m * kP.l[1] + (uint64_t)A + (uint64_t)(C >> 64);
        t0 = (uint64_t)C;
        A = (u128)a.l[2] * bi + t2 + (uint64_t)(A >> 64);
        C = (u128)m * kP.l[2] + (uint64_t)A + (uint64_t)(C >> 64);
        t1 = (uint64_t)C;
        A = (u128)a.l[3] * bi + t3 + (uint64_t)(A >> 64);
        C = (u128)m * kP.l[3] + (uint64_t)A + (uint64_t)(C >> 64);
        t2 = (uint64_t)C;
        A = (u128)a.l[4] * bi + t4 + (uint64_t)(A >> 64);
        C = (u128)m * kP.l[4] + (uint64_t)A + (uint64_t)(C >> 64);
        t3 = (uint64_t)C;
        A = (u128)a.l[5] * bi + t5 + (uint64_t)(A >> 64);
        C = (u128)m * kP.l[5] + (uint64_t)A + (uint64_t)(C >> 64);
        t4 = (uint64_t)C;
        t5 = (uint64_t)(C >> 64) + (uint64_t)(A >> 64);
    }
    const Fp r = {{t0, t1, t2, t3, t4, t5}};
    uint64_t br;
    const Fp s = raw_sub(r, kP, br);
    return br ? r : s;
}
#if defined(__x86_64__)
// The same rounds with mulx and the two independent carry chains of adcx / adox (BMI2 + ADX, checked once at run
// time): t[0..6] += v[0..5] * x per row, first with the row of a, then with the row of p.  ~20 % faster than what the
// compilers make of the portable form; the host tail of a commitment is 400 ... 2000 of these products.
#define KZG_HOST_MAC_ROW(v, x)                                                                                   \
    asm("xorl %k[lo], %k[lo]\n\t"                                                                                 \
        "mulxq 0(%[vp]), %[lo], %[hi]\n\t adcxq %[lo], %[t0]\n\t adoxq %[hi], %[t1]\n\t"                        \
        "mulxq 8(%[vp]), %[lo], %[hi]\n\t adcxq %[lo], %[t1]\n\t adoxq %[hi], %[t2]\n\t"                        \
        "mulxq 16(%[vp]), %[lo], %[hi]\n\t adcxq %[lo], %[t2]\n\t adoxq %[hi], %[t3]\n\t"                       \
        "mulxq 24(%[vp]), %[lo], %[hi]\n\t adcxq %[lo], %[t3]\n\t adoxq %[hi], %[t4]\n\t"                       \
        "mulxq 32(%[vp]), %[lo], %[hi]\n\t adcxq %[lo], %[t4]\n\t adoxq %[hi], %[t5]\n\t"                       \
        "mulxq 40(%[vp]), %[lo], %[hi]\n\t adcxq %[lo], %[t5]\n\t adoxq %[hi], %[t6]\n\t"                       \
        "movl $0, %k[lo]\n\t adcxq %[lo], %[t6]\n\t"                                                             \
        : [t0] "+r"(t0), [t1] "+r"(t1), [t2] "+r"(t2), [t3] "+r"(t3), [t4] "+r"(t4), [t5] "+r"(t5), [t6] "+r"(t6),   \
          [lo] "=&r"(lo), [hi] "=&r"(hi)                                                                         \
        : [vp] "r"(v), "d"(x), "m"(*(const uint64_t(*)[6])(v))                                                    \
        : "cc")
__attribute__((target("bmi2,adx"))) inline Fp mul_adx(const Fp& a, const Fp& b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, lo, hi;
    for (int i = 0; i < 6; ++i) {
        KZG_HOST_MAC_ROW(a.l, b.l[i]);
        const uint64_t m = t0 * kN0;
        KZG_HOST_MAC_ROW(kP.l, m);  // t0 becomes 0: drop it
        t0 = t1; t1 = t2; t2 = t3; t3 = t4; t4 = t5; t5 = t6; t6 = 0;
    }
    const Fp r = {{t0, t1, t2, t3, t4, t5}};
    uint64_t br;
    const Fp s = raw_sub(r, kP, br);
    return br ? r : s;
}
#undef KZG_HOST_MAC_ROW
inline bool cpu_has_adx() {
    static const bool yes = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
    return yes;
}
inline Fp operator*(const Fp& a, const Fp& b) { return cpu_has_adx() ? mul_adx(a, b) : mul_portable(a, b); }
#else
inline Fp operator*(const Fp& a, const Fp& b) { return mul_portable(a, b); }
#endif
inline Fp sqr(const Fp& a) { return a * a; }
inline Fp inv_fermat(const Fp& a) {  // a^(p-2): ~570 products; kept as the cross-check of inv() in tests/host
    Fp e = kP;
    e.l[0] -= 2;
    Fp acc = kOne, base = a;
    for (int i = 0; i < 381; ++i) {
        if ((e.l[i >> 6] >> (i & 63)) & 1) acc = acc * base;
        base = sqr(base);
    }
    return acc;
}
// Inverse by the same division-step routine the device uses (field30_inv.hip.h, host-compiled): the twelve 32-bit
// words of a Montgomery-form element are exactly what fq_from_u32x12 / fq_to_u32x12 exchange.  About a sixth of the
// cost of the Fermat power; it sits on the latency path of every commitment (p1_normalize of the final sum).
inline Fp inv(const Fp& a) {
    if (a.is_zero()) return a;
    uint32_t w[12];
    std::memcpy(w, a.l, sizeof w);
    const kzg::Fq r = kzg::fq_inv(kzg::fq_from_u32x12(w));
    kzg::fq_to_u32x12(r, w);
    Fp out;
    std::memcpy(out.l, w, sizeof w);
    return out;
}
inline Fp from_mont(const Fp& a) {
    Fp one = {{1, 0, 0, 0, 0, 0}};
    return a * one;
}

struct P1 {  // blst_p1 layout: Jacobian, Montgomery, z == 0 <=> infinity
    Fp x, y, z;
    bool is_inf() const { return z.is_zero(); }
};
static_assert(sizeof(P1) == 144, "blst_p1 is 18 x u64");

inline P1 p1_inf() {
    P1 r;
    std::memset(&r, 0, sizeof r);
    return r;
}
inline P1 p1_double(const P1& p) {
    if (p.is_inf() || p.y.is_zero()) return p1_inf();
    Fp A = sqr(p.x), B = sqr(p.y), C = sqr(B);
    Fp t = sqr(p.x + B) - A - C;
    Fp D = t + t;
    Fp E = A + A + A;
    Fp F = sqr(E);
    P1 r;
    r.x = F - D - D;
    Fp C8 = C + C;
    C8 = C8 + C8;
    C8 = C8 + C8;
    r.y = E * (D - r.x) - C8;
    Fp yz = p.y * p.z;
    r.z = yz + yz;
    return r;
}
inline P1 p1_add(const P1& a, const P1& b) {  // complete: handles inf, a == b, a == -b
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    Fp z1z1 = sqr(a.z), z2z2 = sqr(b.z);
    Fp u1 = a.x * z2z2, u2 = b.x * z1z1;
    Fp s1 = a.y * b.z * z2z2, s2 = b.y * a.z * z1z1;
    Fp h = u2 - u1, rr = s2 - s1;
    if (h.is_zero()) return rr.is_zero() ? p1_double(a) : p1_inf();
    Fp hh = sqr(h), hhh = h * hh, v = u1 * hh;
    P1 r;
    r.x = sqr(rr) - hhh - v - v;
    r.y = rr * (v - r.x) - s1 * hhh;
    r.z = a.z * b.z * h;
    return r;
}
inline P1 p1_normalize(const P1& p) {  // affine with z = R (Montgomery one), or all-zero for inf
    if (p.is_inf()) return p1_inf();
    Fp zi = inv(p.z), zi2 = sqr(zi);
    P1 r;
    r.x = p.x * zi2;
    r.y = p.y * zi2 * zi;
    r.z = kOne;
    return r;
}
// One coordinate of a device XYZZ record: 13 signed radix-2^30 digits of a lazily reduced integer v = x * 2^390
// + k p, |v| < 3.5 p (csrc/field30.hip.h)  ->  this file's Montgomery form x * 2^384 mod p.
// v + 4p is positive and below 2^384; (v + 4p) * 2^378 / 2^384 = v * 2^-6 = x * 2^384.
inline Fp fp_from_digits30(const int32_t d[13]) {
    // two's-complement accumulation in seven 64-bit words: big = big * 2^30 + d[i], most significant digit first
    uint64_t w[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 12; i >= 0; --i) {
        for (int k = 6; k > 0; --k) w[k] = (w[k] << 30) | (w[k - 1] >> 34);
        w[0] <<= 30;
        const int64_t add = d[i];
        const uint64_t ext = add < 0 ? ~0ULL : 0ULL;  // sign extension
        u128 c = (u128)w[0] + (uint64_t)add;
        w[0] = (uint64_t)c;
        c >>= 64;
        for (int k = 1; k < 7; ++k) {
            c += (u128)w[k] + ext;
            w[k] = (uint64_t)c;
            c >>= 64;
        }
    }
    static const uint64_t k4P[7] = {0xe7fbfffffffeaaacULL, 0x7aaffffac54ffffeULL, 0x9cc34a83dac3d890ULL, 0x91dd2e13ce144afdULL,
                                    0x2c6e9ed90d2eb35dULL, 0x680447a8e5ff9a69ULL, 0};
    u128 c = 0;
    for (int k = 0; k < 7; ++k) {
        c += (u128)w[k] + k4P[k];
        w[k] = (uint64_t)c;
        c >>= 64;
    }
    Fp v;
    for (int k = 0; k < 6; ++k) v.l[k] = w[k];  // w[6] == 0 by the magnitude bound
    static const Fp k2_378 = {{0, 0, 0, 0, 0, 0x0400000000000000ULL}};
    return mul_wide(v, k2_378);
}
// device XYZZ partial sum (256-byte record: X, Y, ZZ, ZZZ digits at 64-byte steps) -> Jacobian with Z = ZZ:
// X*ZZ, Y*ZZZ, ZZ
inline P1 p1_from_xyzz(const uint64_t* w) {
    const int32_t* d = reinterpret_cast<const int32_t*>(w);
    Fp X = fp_from_digits30(d), Y = fp_from_digits30(d + 16), ZZ = fp_from_digits30(d + 32), ZZZ = fp_from_digits30(d + 48);
    if (ZZ.is_zero()) return p1_inf();
    P1 r;
    r.x = X * ZZ;
    r.y = Y * ZZZ;
    r.z = ZZ;
    return r;
}
// ---- the host tail of an MSM in the device's own coordinates ----------------------------------------------------
// The partial sums arrive as XYZZ records (x = X / ZZ, y = Y / ZZZ).  Adding them as they are costs 12M + 2S per general
// addition against 11M + 5S in Jacobian form, and saves the two products per record of the conversion: ~20 % fewer
// base-field products for the 30 ... 130 additions of a commitment's tail.
struct PX {
    Fp x, y, zz, zzz;
    bool is_inf() const { return zz.is_zero(); }
};
inline PX px_inf() {
    PX r;
    std::memset(&r, 0, sizeof r);
    return r;
}
inline PX px_from_record(const uint64_t* w) {
    const int32_t* d = reinterpret_cast<const int32_t*>(w);
    PX r;
    r.zz = fp_from_digits30(d + 32);
    if (r.zz.is_zero()) return px_inf();
    r.x = fp_from_digits30(d);
    r.y = fp_from_digits30(d + 16);
    r.zzz = fp_from_digits30(d + 48);
    return r;
}
inline PX px_double(const PX& p) {  // dbl-2008-s-1 (a = 0)
    if (p.is_inf() || p.y.is_zero()) return px_inf();
    const Fp u = p.y + p.y, v = sqr(u), w = u * v, s = p.x * v;
    const Fp xx = sqr(p.x), m = xx + xx + xx;
    PX r;
    r.x = sqr(m) - s - s;
    r.y = m * (s - r.x) - w * p.y;
    r.zz = v * p.zz;
    r.zzz = w * p.zzz;
    return r;
}
inline PX px_add(const PX& a, const PX& b) {  // add-2008-s, complete: handles inf, a == b, a == -b
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    const Fp u1 = a.x * b.zz, u2 = b.x * a.zz, s1 = a.y * b.zzz, s2 = b.y * a.zzz;
    const Fp pp = u2 - u1, rr = s2 - s1;
    if (pp.is_zero()) return rr.is_zero() ? px_double(a) : px_inf();
    const Fp p2 = sqr(pp), p3 = pp * p2, q = u1 * p2;
    PX r;
    r.x = sqr(rr) - p3 - q - q;
    r.y = rr * (q - r.x) - s1 * p3;
    r.zz = a.zz * b.zz * p2;
    r.zzz = a.zzz * b.zzz * p3;
    return r;
}
// XYZZ -> Jacobian without an inversion: Z = ZZ gives Z^2 = ZZ^2, Z^3 = ZZ^3 = ZZZ^2, so (X ZZ, Y ZZZ, ZZ).
// The un-normalised partial sum a device of a range-split context hands to the exchange (multi.hip).
inline P1 px_to_jacobian(const PX& p) {
    if (p.is_inf()) return p1_inf();
    P1 r;
    r.x = p.x * p.zz;
    r.y = p.y * p.zzz;
    r.z = p.zz;
    return r;
}
// normalises k Jacobian points with ONE inversion (Montgomery's trick over the Z coordinates; infinities skipped)
inline void p1_normalize_many(P1* pts, size_t k) {
    std::vector<Fp> prefix(k);
    Fp run = kOne;
    for (size_t i = 0; i < k; i++) {
        prefix[i] = run;
        if (!pts[i].is_inf()) run = run * pts[i].z;
    }
    Fp inv_run = inv(run);
    for (size_t i = k; i-- > 0;) {
        if (pts[i].is_inf()) {
            pts[i] = p1_inf();
            continue;
        }
        const Fp zi = inv_run * prefix[i], zi2 = sqr(zi);
        inv_run = inv_run * pts[i].z;
        pts[i].x = pts[i].x * zi2;
        pts[i].y = pts[i].y * zi2 * zi;
        pts[i].z = kOne;
    }
}
inline P1 px_normalize(const PX& p) {  // affine with z = R, or all-zero for inf: one inversion for both denominators
    if (p.is_inf()) return p1_inf();
    const Fp i = inv(p.zz * p.zzz);
    P1 r;
    r.x = p.x * (i * p.zzz);
    r.y = p.y * (i * p.zz);
    r.z = kOne;
    return r;
}

inline void p1_compress(uint8_t out[48], const P1& p) {  // ZCash encoding (reference src/curves.rs:99-110)
    if (p.is_inf()) {
        std::memset(out, 0, 48);
        out[0] = 0xC0;
        return;
    }
    P1 a = p1_normalize(p);
    Fp x = from_mont(a.x), y = from_mont(a.y);
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 8; ++k) out[47 - (8 * i + k)] = (uint8_t)(x.l[i] >> (8 * k));
    out[0] |= 0x80;
    if (!(y == kHalf) && geq(y, kHalf)) out[0] |= 0x20;
}

// inverse of p1_compress: what the reference's `Deserialize for G1Point` does through blst_p1_uncompress
// (src/curves.rs:112-183).  On-curve check only (as blst), y = (x^3 + 4)^((p+1)/4) since p = 3 mod 4.
inline bool p1_uncompress(P1& out, const uint8_t in[48]) {
    if (!(in[0] & 0x80)) return false;          // compressed form only
    if (in[0] & 0x40) {                          // infinity: all other bits must be clear
        for (int i = 0; i < 48; ++i)
            if ((i == 0 ? (in[0] & 0x3F) : in[i]) != 0) return false;
        out = p1_inf();
        return true;
    }
    Fp x;
    std::memset(&x, 0, sizeof x);
    for (int i = 0; i < 48; ++i) {
        uint8_t b = i == 0 ? (uint8_t)(in[0] & 0x1F) : in[i];
        x.l[(47 - i) >> 3] |= (uint64_t)b << (8 * ((47 - i) & 7));
    }
    if (geq(x, kP)) return false;
    static const Fp kR2 = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                            0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};
    static const Fp kB = {{0xaa270000000cfff3ULL, 0x53cc0032fc34000aULL, 0x478fe97a6b0a807fULL,
                           0xb1d37ebee6ba24d7ULL, 0x8ec9733bbf78ab2fULL, 0x09d645513d83de7eULL}};  // 4
    static const Fp kSqrtE = {{0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL,
                               0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL}};  // (p+1)/4
    Fp xm = x * kR2;
    Fp y2 = sqr(xm) * xm + kB;
    Fp acc = kOne, base = y2;
    for (int i = 0; i < 381; ++i) {
        if ((kSqrtE.l[i >> 6] >> (i & 63)) & 1) acc = acc * base;
        base = sqr(base);
    }
    if (!(sqr(acc) == y2)) return false;         // not on the curve
    Fp yc = from_mont(acc);
    bool larger = !(yc == kHalf) && geq(yc, kHalf);
    if (larger != ((in[0] & 0x20) != 0)) acc = neg(acc);
    out.x = xm;
    out.y = acc;
    out.z = kOne;
    return true;
}

}  // namespace kzg_host
