// field30_inv.hip.h -- modular inversion in Fp by Bernstein-Yang "safegcd" division steps on signed radix-2^30
// digits, branch-free (every lane of a wavefront runs the same instruction stream).
//
// Why not Fermat: a^(p-2) costs ~460 field products per lane; batched affine additions (msm_accum.hip) invert one
// product of denominators per lane per batch, so the inversion has to cost a few dozen products, not hundreds.
// 30 batches of 30 division steps on the low words of (f, g) give 2x2 transition matrices with 30-bit entries;
// each matrix is applied to the full-width f, g and to the Bezout coefficients d, e (kept mod p) with
// v_mad_i64_i32 chains -- the same instruction the multiplier lives on.  900 steps cover the 878 that
// 381-bit inputs can need ((45907 n + 26313) / 19929 with n = 381; Bernstein-Yang 2019, Pornin's/Wuille's
// half-delta variant).  Cost: ~30 x (30 x 17 simple instructions + 130 multiply-adds) = 35-40 field products.
//
// Number format inside this file ("s30"): 13 words, words 0..11 in [0, 2^30), word 12 signed.
#pragma once
#include "field30.hip.h"

namespace kzg {

struct S30 {
    int32_t v[kQ];
};

// p as s30 words (unsigned digits)
KZG_HD constexpr int32_t s30_p(int i) {
    constexpr int32_t PW[13] = {
#include "field30_p_u30.inc"
    };
    return PW[i];
}

// 30 division steps on the low words; returns the new zeta and the matrix t = (u, v, q, r) with
// t * (f, g) = 2^30 * (f', g')
KZG_HD int32_t s30_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t t[4]) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t f = f0, g = g0;
#pragma unroll 1
    for (int i = 0; i < 30; i++) {
        uint32_t c1 = (uint32_t)(zeta >> 31);  // all ones when zeta < 0
        const uint32_t c2 = 0u - (g & 1u);     // all ones when g is odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // (f, u, v) negated when zeta < 0
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;  // swap only when zeta < 0 and g odd
        zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t[0] = (int32_t)u;
    t[1] = (int32_t)v;
    t[2] = (int32_t)q;
    t[3] = (int32_t)r;
    return zeta;
}

// (f, g) <- t * (f, g) / 2^30  (exact)
KZG_HD void s30_update_fg(S30& f, S30& g, const int32_t t[4]) {
    const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
    int64_t cf = u * f.v[0] + v * g.v[0];
    int64_t cg = q * f.v[0] + r * g.v[0];
    cf >>= kQBits;
    cg >>= kQBits;
#pragma unroll
    for (int i = 1; i < kQ; i++) {
        cf += u * f.v[i] + v * g.v[i];
        cg += q * f.v[i] + r * g.v[i];
        f.v[i - 1] = (int32_t)cf & kQMask;
        g.v[i - 1] = (int32_t)cg & kQMask;
        cf >>= kQBits;
        cg >>= kQBits;
    }
    f.v[kQ - 1] = (int32_t)cf;
    g.v[kQ - 1] = (int32_t)cg;
}

// (d, e) <- t * (d, e) / 2^30 mod p, both kept in (-2p, p)
KZG_HD void s30_update_de(S30& d, S30& e, const int32_t t[4]) {
    const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
    const int32_t sd = d.v[kQ - 1] >> 31, se = e.v[kQ - 1] >> 31;  // sign masks
    int32_t md = (u & sd) + (v & se);
    int32_t me = (q & sd) + (r & se);
    int64_t cd = (int64_t)u * d.v[0] + (int64_t)v * e.v[0];
    int64_t ce = (int64_t)q * d.v[0] + (int64_t)r * e.v[0];
    // make the low 30 bits of cd + p*md vanish: md -= (p^-1 * cd + md) mod 2^30
    const uint32_t pinv = (1u << kQBits) - kQN0;  // +p^-1 mod 2^30
    md -= (int32_t)((pinv * (uint32_t)cd + (uint32_t)md) & (uint32_t)kQMask);
    me -= (int32_t)((pinv * (uint32_t)ce + (uint32_t)me) & (uint32_t)kQMask);
    cd += (int64_t)s30_p(0) * md;
    ce += (int64_t)s30_p(0) * me;
    cd >>= kQBits;
    ce >>= kQBits;
#pragma unroll
    for (int i = 1; i < kQ; i++) {
        cd += (int64_t)u * d.v[i] + (int64_t)v * e.v[i] + (int64_t)s30_p(i) * md;
        ce += (int64_t)q * d.v[i] + (int64_t)r * e.v[i] + (int64_t)s30_p(i) * me;
        d.v[i - 1] = (int32_t)cd & kQMask;
        e.v[i - 1] = (int32_t)ce & kQMask;
        cd >>= kQBits;
        ce >>= kQBits;
    }
    d.v[kQ - 1] = (int32_t)cd;
    e.v[kQ - 1] = (int32_t)ce;
}

// integer in [0, p) as s30 words from a lazy signed-digit value
KZG_HD S30 s30_from_fq(const Fq& a_in) {
    // reduce below 0.62 p in magnitude, make the digits canonical, add p when negative, then unsigned digits
    Fq a = fq_canon_digits(fq_mul(a_in, fq_one()));
    int32_t top = a.d[kQ - 1];
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < kQ - 1; i++) s = a.d[i] != 0 ? a.d[i] : s;
    const bool negative = top < 0 || (top == 0 && s < 0);
    S30 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < kQ - 1; i++) {
        int32_t w = a.d[i] + (negative ? fq_pd(i) : 0) + c;
        c = w >> kQBits;
        r.v[i] = w & kQMask;
    }
    r.v[kQ - 1] = a.d[kQ - 1] + (negative ? fq_pd(kQ - 1) : 0) + c;
    return r;
}

// a^-1 in the same Montgomery form: for the representative v = x * 2^390 returns a value congruent to
// x^-1 * 2^390, digits as fq_mul leaves them.  a must not be 0 mod p (returns 0 then).
KZG_HD Fq fq_inv(const Fq& a) {
    S30 d, e, f, g = s30_from_fq(a);
#pragma unroll
    for (int i = 0; i < kQ; i++) {
        d.v[i] = 0;
        e.v[i] = 0;
        f.v[i] = s30_p(i);
    }
    e.v[0] = 1;
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 30; it++) {
        int32_t t[4];
        zeta = s30_divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        s30_update_de(d, e, t);
        s30_update_fg(f, g, t);
    }
    // g = 0, f = +-1 (or +-p when the input was 0); the inverse is sign(f) * d, d in (-2p, p)
    const int32_t fneg = f.v[kQ - 1] >> 31;  // all ones when f = -1
    Fq r;
#pragma unroll
    for (int i = 0; i < kQ; i++) r.d[i] = (d.v[i] ^ fneg) - fneg;
    // r is an integer congruent to v^-1 with |r| < 2p and words of magnitude < 2^30: one carry pass, then
    // v^-1 = x^-1 2^-390  ->  x^-1 2^390 needs the factor 2^780: as a Montgomery product with 2^1170 mod p
    r = fq_norm(r);
    return fq_mul(r, fq_const_2_1170());
}

}  // namespace kzg
