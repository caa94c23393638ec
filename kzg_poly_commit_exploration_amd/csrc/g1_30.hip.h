// g1_30.hip.h -- BLS12-381 G1 group law on the signed radix-2^30 field of field30.hip.h.
//
// Same formulas and the same complete exceptional handling as g1.hip.h (XYZZ accumulator += affine point,
// madd-2008-s: 8M + 2S), restated for lazily reduced signed values: sums and differences are digit-wise and get
// ONE parallel carry pass per expression; nothing is ever conditionally reduced mod p.  Replaces what the reference
// obtains from blst_p1_add_or_double / blst_p1_mult inside Polynomial::commit (src/polynomial.rs:207-212,
// src/curves.rs:79-96).
//
// Magnitudes (p-multiples; every product is < 0.62 p + |a||b| / 2^390, and p / 2^390 < 0.0016):
//   X < 2.6 p, Y < 1.3 p, ZZ, ZZZ < 0.7 p;   P = U2 - X < 3.3 p, R = S2 - Y < 2 p  -> zero tests valid below 3.5 p.
#pragma once
#include "field30.hip.h"

namespace kzg {

// scheduling barrier between the products of the group law: the compiler otherwise interleaves the ten
// independent-looking multiplications and the live ranges of their operands push the kernel over 2 waves/SIMD
#if defined(__HIP_DEVICE_COMPILE__)
#define KZG_SB30() __builtin_amdgcn_sched_barrier(0)
#else
#define KZG_SB30() ((void)0)
#endif

struct Affine30 {  // Montgomery (R' = 2^390) x, y; all digits zero = the point at infinity
    Fq x, y;
};

struct XYZZ30 {
    Fq X, Y, ZZ, ZZZ;
};

KZG_HD Fq fq_one() {  // 2^390 mod p, balanced digits (tools/gen_field30_constants.py)
    Fq c;
    constexpr int32_t V[13] = {
#include "field30_one.inc"
    };
#pragma unroll
    for (int i = 0; i < kQ; i++) c.d[i] = V[i];
    return c;
}

KZG_HD bool fq_all_zero(const Fq& a) {
    int32_t o = 0;
#pragma unroll
    for (int i = 0; i < kQ; i++) o |= a.d[i];
    return o == 0;
}

KZG_HD XYZZ30 xyzz30_inf() {
    XYZZ30 r;
    r.X = fq_zero();
    r.Y = fq_zero();
    r.ZZ = fq_zero();
    r.ZZZ = fq_zero();
    return r;
}
KZG_HD bool xyzz30_is_inf(const XYZZ30& a) { return fq_all_zero(a.ZZ); }  // infinity is only ever written as exact zeros

// 2 * a (dbl-2008-s-1).  Rare path of the additions (equal operands) and the table construction.
#ifdef __HIPCC__
__device__ __noinline__
#else
inline
#endif
    void
    xyzz30_dbl_inplace(XYZZ30& a) {
    if (xyzz30_is_inf(a) || fq_is_zero(a.Y)) {
        a = xyzz30_inf();
        return;
    }
    const Fq U = fq_norm(fq_add_raw(a.Y, a.Y));
    const Fq V = fq_sqr(U);
    const Fq W = fq_mul(U, V);
    const Fq S = fq_mul(a.X, V);
    Fq M = fq_sqr(a.X);
    M = fq_norm(fq_add_raw(fq_add_raw(M, M), M));
    const Fq X3 = fq_norm(fq_sub_raw(fq_sub_raw(fq_sqr(M), S), S));
    const Fq Y3 = fq_norm(fq_sub_raw(fq_mul(M, fq_sub_raw(S, X3)), fq_mul(W, a.Y)));
    a.ZZ = fq_mul(V, a.ZZ);
    a.ZZZ = fq_mul(W, a.ZZZ);
    a.X = X3;
    a.Y = Y3;
}

// acc += p (affine), p negated when `neg`; complete.
KZG_HD void xyzz30_madd(XYZZ30& acc, const Affine30& p_in, bool neg) {
    if (fq_all_zero(p_in.x) && fq_all_zero(p_in.y)) return;
    const Fq py = fq_cneg(p_in.y, neg);
    if (xyzz30_is_inf(acc)) {
        acc.X = p_in.x;
        acc.Y = py;
        acc.ZZ = fq_one();
        acc.ZZZ = fq_one();
        return;
    }
    const Fq P = fq_norm(fq_sub_raw(fq_mul(p_in.x, acc.ZZ), acc.X));  // U2 - X1
    KZG_SB30();
    const Fq R = fq_norm(fq_sub_raw(fq_mul(py, acc.ZZZ), acc.Y));     // S2 - Y1
    KZG_SB30();
    if (fq_is_zero(P)) {
        if (fq_is_zero(R)) xyzz30_dbl_inplace(acc);  // acc == p as group elements
        else acc = xyzz30_inf();
        return;
    }
    const Fq PP = fq_sqr(P);
    KZG_SB30();
    acc.ZZ = fq_mul(acc.ZZ, PP);
    KZG_SB30();
    const Fq Q = fq_mul(acc.X, PP);
    KZG_SB30();
    const Fq PPP = fq_mul(P, PP);
    KZG_SB30();
    acc.ZZZ = fq_mul(acc.ZZZ, PPP);
    KZG_SB30();
    const Fq YP = fq_mul(acc.Y, PPP);
    KZG_SB30();
    const Fq X3 = fq_norm_wide(fq_sub_raw(fq_sub_raw(fq_sqr(R), PPP), fq_add_raw(Q, Q)));  // digits up to 2^31
    KZG_SB30();
    acc.Y = fq_norm(fq_sub_raw(fq_mul(R, fq_sub_raw(Q, X3)), YP));
    KZG_SB30();
    acc.X = X3;
}

}  // namespace kzg
