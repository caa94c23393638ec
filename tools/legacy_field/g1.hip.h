// g1.hip.h -- BLS12-381 G1 group law on the device (y^2 = x^3 + 4 over Fp).
//
// Replaces what the reference obtains from blst_p1_add_or_double (reference src/curves.rs:79-85)
// and blst_p1_mult (src/curves.rs:90-96) inside Polynomial::commit (src/polynomial.rs:207-212).
// Accumulators are extended Jacobian "XYZZ" (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; ZZ = 0 <=> infinity):
// the mixed addition with an affine SRS point costs 8M + 2S and the exceptional cases
// (accumulator at infinity, equal points, opposite points) are handled explicitly -- they do
// occur (SRS[0] = G, repeated coefficients) and the output must be bit-exact, not probable.
#pragma once
#include "field_fips.hip.h"

namespace kzg {

#ifndef KZG_SB
#define KZG_SB() __builtin_amdgcn_sched_barrier(0)
#endif

// Fp products of the group law go through fmul/fsqr.  With KZG_MUL_CALL they are real function
// calls (one copy of the ~1000-instruction multiplier in the instruction cache instead of ten per
// addition); otherwise they are inlined.  Chosen per kernel from measurements.
// KZG_LAZY_FP (set for every device translation unit by the Makefile, or not at all): coordinates live
// in [0, 2p) and the multiplier skips its final conditional subtraction (field.hip.h).
#ifdef KZG_LAZY_FP
#define KZG_FMUL_IMPL(a, b) fe_mul_fips<FpParams, true>(a, b)
#define KZG_FSQR_LAZY true
KZG_DEV Fp fadd(const Fp& a, const Fp& b) { return fp_add_lz(a, b); }
KZG_DEV Fp fsub(const Fp& a, const Fp& b) { return fp_sub_lz(a, b); }
KZG_DEV Fp fneg(const Fp& a) { return fp_neg_lz(a); }
KZG_DEV bool fzero(const Fp& a) { return fp_is_zero_lz(a); }
KZG_DEV Fp fcanon(const Fp& a) { return fp_canon(a); }
#else
#define KZG_FMUL_IMPL(a, b) fe_mul_fips<FpParams, false>(a, b)
#define KZG_FSQR_LAZY false
KZG_DEV Fp fadd(const Fp& a, const Fp& b) { return fe_add(a, b); }
KZG_DEV Fp fsub(const Fp& a, const Fp& b) { return fe_sub(a, b); }
KZG_DEV Fp fneg(const Fp& a) { return fe_neg(a); }
KZG_DEV bool fzero(const Fp& a) { return a.is_zero(); }
KZG_DEV Fp fcanon(const Fp& a) { return a; }
#endif
// KZG_FIPS_SQR: dedicated squaring (89 instead of 144 multiply-adds in the a*a half)
#ifdef KZG_FIPS_SQR
#define KZG_FSQR_IMPL(a) fe_sqr_fips<FpParams, KZG_FSQR_LAZY>(a)
#else
#define KZG_FSQR_IMPL(a) KZG_FMUL_IMPL(a, a)
#endif
#ifdef KZG_MUL_CALL
static __device__ __noinline__ Fp fmul(Fp a, Fp b) { return KZG_FMUL_IMPL(a, b); }
static __device__ __noinline__ Fp fsqr(Fp a) { return KZG_FSQR_IMPL(a); }
#else
KZG_DEV Fp fmul(const Fp& a, const Fp& b) { return KZG_FMUL_IMPL(a, b); }
KZG_DEV Fp fsqr(const Fp& a) { return KZG_FSQR_IMPL(a); }
#endif
KZG_DEV Fp fdbl(const Fp& a) { return fadd(a, a); }

struct Affine {  // Montgomery x, y;  (0, 0) encodes the point at infinity (it is not on the curve)
    Fp x, y;
    KZG_DEV bool is_inf() const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) o |= x.l[i] | y.l[i];
        return o == 0;
    }
};

struct XYZZ {
    Fp X, Y, ZZ, ZZZ;
    KZG_DEV static XYZZ inf() {
        XYZZ r;
        r.X = Fp::zero();
        r.Y = Fp::zero();
        r.ZZ = Fp::zero();
        r.ZZZ = Fp::zero();
        return r;
    }
    KZG_DEV bool is_inf() const { return fzero(ZZ); }
};

KZG_DEV XYZZ xyzz_from_affine(const Affine& p) {
    XYZZ r;
    if (p.is_inf()) return XYZZ::inf();
    r.X = p.x;
    r.Y = p.y;
    r.ZZ = Fp::one();
    r.ZZZ = Fp::one();
    return r;
}

// 2 * a (dbl-2008-s-1), fully unrolled: for kernels whose main operation is doubling.
KZG_DEV XYZZ xyzz_dbl(const XYZZ& a) {
    if (a.is_inf() || fzero(a.Y)) return XYZZ::inf();
    XYZZ r;
    Fp U = fdbl(a.Y);
    Fp V = fsqr(U);
    KZG_SB();
    Fp W = fmul(U, V);
    KZG_SB();
    Fp S = fmul(a.X, V);
    KZG_SB();
    r.ZZ = fmul(V, a.ZZ);
    KZG_SB();
    Fp M = fsqr(a.X);
    KZG_SB();
    M = fadd(fdbl(M), M);
    r.X = fsub(fsub(fsqr(M), S), S);
    KZG_SB();
    r.ZZZ = fmul(W, a.ZZZ);
    KZG_SB();
    Fp WY = fmul(W, a.Y);
    KZG_SB();
    r.Y = fsub(fmul(M, fsub(S, r.X)), WY);
    KZG_SB();
    return r;
}

// 2 * a on the rolled routines (exceptional branch of the additions: equal operands).
// io = {X, Y, ZZ, ZZZ} as 4 x 12 words in private memory, overwritten with the double.
static __device__ __noinline__ void xyzz_dbl_rare(u32* io) {
    u32 *X = io, *Y = io + 12, *ZZ = io + 24, *ZZZ = io + 36;
    u32 U[12], V[12], W[12], S[12], M[12], T[12];
    fp_rolled_add(U, Y, Y);
    fp_rolled_mul(V, U, U);
    fp_rolled_mul(W, U, V);
    fp_rolled_mul(S, X, V);
    fp_rolled_mul(M, X, X);
    fp_rolled_add(T, M, M);
    fp_rolled_add(M, T, M);
    fp_rolled_mul(T, M, M);
    fp_rolled_sub(T, T, S);
    fp_rolled_sub(T, T, S);  // X3
    fp_rolled_mul(ZZ, V, ZZ);
    fp_rolled_mul(ZZZ, W, ZZZ);
    fp_rolled_mul(W, W, Y);  // W * Y1
    fp_rolled_sub(S, S, T);
    fp_rolled_mul(S, M, S);
    fp_rolled_sub(Y, S, W);
#pragma unroll 1
    for (int i = 0; i < 12; i++) X[i] = T[i];
}

// acc = 2 * acc through the rare path (caller has already excluded infinity / Y = 0 handled here)
KZG_DEV void xyzz_dbl_inplace_rare(XYZZ& acc) {
    if (acc.is_inf() || fzero(acc.Y)) {
        acc = XYZZ::inf();
        return;
    }
    u32 io[48];
    const Fp cx = fcanon(acc.X), cy = fcanon(acc.Y), czz = fcanon(acc.ZZ), czzz = fcanon(acc.ZZZ);  // rolled code is strict mod p
#pragma unroll
    for (int i = 0; i < 12; i++) {
        io[i] = cx.l[i];
        io[12 + i] = cy.l[i];
        io[24 + i] = czz.l[i];
        io[36 + i] = czzz.l[i];
    }
    xyzz_dbl_rare(io);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        acc.X.l[i] = io[i];
        acc.Y.l[i] = io[12 + i];
        acc.ZZ.l[i] = io[24 + i];
        acc.ZZZ.l[i] = io[36 + i];
    }
}

// acc += p (affine), p negated when `neg`.  madd-2008-s with complete exceptional handling.
// Statement order and the scheduling barriers keep at most ~8 field elements live (the compiler
// otherwise interleaves the ten independent-looking products and spills).
KZG_DEV void xyzz_madd(XYZZ& acc, const Affine& p_in, bool neg) {
    if (p_in.is_inf()) return;
    Fp py = neg ? fneg(p_in.y) : p_in.y;
    if (acc.is_inf()) {
        acc.X = p_in.x;
        acc.Y = py;
        acc.ZZ = Fp::one();
        acc.ZZZ = Fp::one();
        return;
    }
    Fp P = fsub(fmul(p_in.x, acc.ZZ), acc.X);  // U2 - X1
    KZG_SB();
    Fp R = fsub(fmul(py, acc.ZZZ), acc.Y);  // S2 - Y1
    KZG_SB();
    if (fzero(P)) {
        if (fzero(R)) {
            xyzz_dbl_inplace_rare(acc);  // acc == p as group elements: 2*acc
        } else {
            acc = XYZZ::inf();
        }
        return;
    }
    Fp PP = fsqr(P);
    KZG_SB();
    acc.ZZ = fmul(acc.ZZ, PP);
    KZG_SB();
    Fp Q = fmul(acc.X, PP);
    KZG_SB();
    Fp PPP = fmul(P, PP);
    KZG_SB();
    acc.ZZZ = fmul(acc.ZZZ, PPP);
    KZG_SB();
    Fp YP = fmul(acc.Y, PPP);
    KZG_SB();
    Fp X3 = fsub(fsub(fsqr(R), PPP), fdbl(Q));
    KZG_SB();
    acc.Y = fsub(fmul(R, fsub(Q, X3)), YP);
    acc.X = X3;
    KZG_SB();
}

// acc += b : add-2008-s with complete exceptional handling.
KZG_DEV void xyzz_add(XYZZ& acc, const XYZZ& b) {
    if (b.is_inf()) return;
    if (acc.is_inf()) {
        acc = b;
        return;
    }
    Fp U1 = fmul(acc.X, b.ZZ);
    KZG_SB();
    Fp P = fsub(fmul(b.X, acc.ZZ), U1);
    KZG_SB();
    Fp S1 = fmul(acc.Y, b.ZZZ);
    KZG_SB();
    Fp R = fsub(fmul(b.Y, acc.ZZZ), S1);
    KZG_SB();
    if (fzero(P)) {
        if (fzero(R)) {
            // equal operands: frequent in the running-sum reduction when buckets are empty
#ifdef KZG_FAST_DBL_IN_ADD
            acc = xyzz_dbl(acc);
#else
            xyzz_dbl_inplace_rare(acc);
#endif
        } else {
            acc = XYZZ::inf();
        }
        return;
    }
    Fp PP = fsqr(P);
    KZG_SB();
    acc.ZZ = fmul(fmul(acc.ZZ, b.ZZ), PP);
    KZG_SB();
    Fp Q = fmul(U1, PP);
    KZG_SB();
    Fp PPP = fmul(P, PP);
    KZG_SB();
    acc.ZZZ = fmul(fmul(acc.ZZZ, b.ZZZ), PPP);
    KZG_SB();
    Fp YP = fmul(S1, PPP);
    KZG_SB();
    Fp X3 = fsub(fsub(fsqr(R), PPP), fdbl(Q));
    KZG_SB();
    acc.Y = fsub(fmul(R, fsub(Q, X3)), YP);
    acc.X = X3;
    KZG_SB();
}

// Jacobian (X, Y, Z) as stored in blst_p1 -> XYZZ (no inversion): ZZ = Z^2, ZZZ = Z^3.
KZG_DEV XYZZ xyzz_from_jacobian(const Fp& X, const Fp& Y, const Fp& Z) {
    XYZZ r;
    if (fzero(Z)) return XYZZ::inf();
    r.X = X;
    r.Y = Y;
    r.ZZ = fsqr(Z);
    r.ZZZ = fmul(r.ZZ, Z);
    return r;
}

// XYZZ -> Jacobian with Z' = ZZ*ZZZ... not needed: with the hidden z (ZZ = z^2, ZZZ = z^3),
// (X*ZZ, Y*ZZZ, ZZ... ) is NOT Jacobian; use: Z' = ZZZ/ZZ is unavailable without inversion, so take
// Z' = ZZ * ZZZ = z^5:  X' = x z'^2 = X * ZZ^4 ... cheaper: Z' = ZZ (= z^2): X' = x*ZZ^2 = X*ZZ,
// Y' = y*ZZ^3 = Y*ZZZ.  Three multiplications, no inversion.
KZG_DEV void xyzz_to_jacobian(const XYZZ& a, Fp& X, Fp& Y, Fp& Z) {
    if (a.is_inf()) {
        X = Fp::zero();
        Y = Fp::zero();
        Z = Fp::zero();
        return;
    }
    X = fmul(a.X, a.ZZ);
    Y = fmul(a.Y, a.ZZZ);
    Z = a.ZZ;
}

}  // namespace kzg
