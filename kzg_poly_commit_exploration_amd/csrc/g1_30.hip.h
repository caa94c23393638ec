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
// (KZG_G1_30_NO_SB: latency-bound kernels -- one wave per SIMD, 512 registers to spend -- want the opposite)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_G1_30_NO_SB)
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

// the Montgomery one for the (rare) "accumulator was at infinity" branch: a real call on the device, so that the
// compiler does not materialise its 13 digits in registers, twice, in front of the branch on every addition
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __noinline__ Fq fq_one_cold() { return fq_one(); }
#else
KZG_HD Fq fq_one_cold() { return fq_one(); }
#endif

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

// 2 * a (dbl-2008-s-1)
KZG_HD void xyzz30_dbl_body(XYZZ30& a) {
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
// the same as a real call: the rare equal-operands branch of the accumulation kernel must not set the register
// budget of its hot loop (KZG_G1_30_INLINE_DBL: kernels that prefer no private memory inline it instead)
// The accumulator itself must never have its address taken (it would live in private memory for the whole kernel):
// the call works on a copy.
#if defined(__HIPCC__) && !defined(KZG_G1_30_INLINE_DBL)
static __device__ __noinline__ void xyzz30_dbl_call(XYZZ30* io) { xyzz30_dbl_body(*io); }
__device__ __forceinline__ void xyzz30_dbl_inplace(XYZZ30& a) {
    XYZZ30 tmp = a;
    xyzz30_dbl_call(&tmp);
    a = tmp;
}
#else
KZG_HD void xyzz30_dbl_inplace(XYZZ30& a) { xyzz30_dbl_body(a); }
#endif

// acc += p (affine), p negated when `neg`; complete.  Two halves so that a caller can reuse the registers of the
// point between them (the accumulation kernel issues the gather of its next point there):
//   xyzz30_madd_head  consumes the point: returns false when the addition is already complete (a trivial or
//                     exceptional case), true with P = U2 - X1 and R = S2 - Y1 otherwise;
//   xyzz30_madd_tail  the remaining 6M + 2S on (acc, P, R); the last two products share one reduction.
KZG_HD bool xyzz30_madd_head(XYZZ30& acc, const Affine30& p_in, bool neg, Fq& P, Fq& R) {
    if (fq_all_zero(p_in.x) && fq_all_zero(p_in.y)) return false;
    const Fq py = fq_cneg(p_in.y, neg);
    if (xyzz30_is_inf(acc)) {
        acc.X = p_in.x;
        acc.Y = py;
        acc.ZZ = fq_one_cold();
        acc.ZZZ = acc.ZZ;
        return false;
    }
    P = fq_norm(fq_sub_raw(fq_mul(p_in.x, acc.ZZ), acc.X));  // U2 - X1
    KZG_SB30();
    R = fq_norm(fq_sub_raw(fq_mul(py, acc.ZZZ), acc.Y));     // S2 - Y1
    KZG_SB30();
    if (fq_is_zero(P)) {
        if (fq_is_zero(R)) xyzz30_dbl_inplace(acc);  // acc == p as group elements
        else acc = xyzz30_inf();
        return false;
    }
    return true;
}
KZG_HD void xyzz30_madd_tail(XYZZ30& acc, const Fq& P, const Fq& R) {
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
    const Fq X3 = fq_norm_wide(fq_sub_raw(fq_sub_raw(fq_sqr(R), PPP), fq_add_raw(Q, Q)));  // digits up to 2^31
    KZG_SB30();
#ifdef KZG_NO_MUL_SUB  // A/B: two products, two reductions (round 2's first form)
    acc.Y = fq_norm(fq_sub_raw(fq_mul(R, fq_sub_raw(Q, X3)), fq_mul(acc.Y, PPP)));
#else
    acc.Y = fq_mul_sub(R, fq_norm(fq_sub_raw(Q, X3)), acc.Y, PPP);  // R (Q - X3) - Y1 PPP, one reduction
#endif
    KZG_SB30();
    acc.X = X3;
}
KZG_HD void xyzz30_madd(XYZZ30& acc, const Affine30& p_in, bool neg) {
    Fq P, R;
    if (xyzz30_madd_head(acc, p_in, neg, P, R)) xyzz30_madd_tail(acc, P, R);
}

// acc += b (add-2008-s: 12M + 2S, the last two products sharing one reduction), complete
KZG_HD void xyzz30_add(XYZZ30& acc, const XYZZ30& b) {
    if (xyzz30_is_inf(b)) return;
    if (xyzz30_is_inf(acc)) {
        acc = b;
        return;
    }
    const Fq U1 = fq_mul(acc.X, b.ZZ);
    KZG_SB30();
    const Fq P = fq_norm(fq_sub_raw(fq_mul(b.X, acc.ZZ), U1));
    KZG_SB30();
    const Fq S1 = fq_mul(acc.Y, b.ZZZ);
    KZG_SB30();
    const Fq R = fq_norm(fq_sub_raw(fq_mul(b.Y, acc.ZZZ), S1));
    KZG_SB30();
    if (fq_is_zero(P)) {
        if (fq_is_zero(R)) xyzz30_dbl_inplace(acc);
        else acc = xyzz30_inf();
        return;
    }
    const Fq PP = fq_sqr(P);
    KZG_SB30();
    acc.ZZ = fq_mul(fq_mul(acc.ZZ, b.ZZ), PP);
    KZG_SB30();
    const Fq Q = fq_mul(U1, PP);
    KZG_SB30();
    const Fq PPP = fq_mul(P, PP);
    KZG_SB30();
    acc.ZZZ = fq_mul(fq_mul(acc.ZZZ, b.ZZZ), PPP);
    KZG_SB30();
    const Fq X3 = fq_norm_wide(fq_sub_raw(fq_sub_raw(fq_sqr(R), PPP), fq_add_raw(Q, Q)));
    KZG_SB30();
    acc.Y = fq_mul_sub(R, fq_norm(fq_sub_raw(Q, X3)), S1, PPP);  // R (Q - X3) - S1 PPP, one reduction
    acc.X = X3;
}

// ---- affine pair additions with a shared inversion (the accumulation kernel's front end) ----------------------------
// Two table points that fall into the same bucket are first added in AFFINE coordinates: lambda = num / den,
// x3 = lambda^2 - xa - xb, y3 = lambda (xa - x3) - ya -- 2M + 1S once 1/den is known, and the sum enters the bucket's
// XYZZ accumulator with ONE mixed addition instead of two.  The denominators of all the pairs of a lane are inverted
// together (Montgomery's trick: 3 products per pair + one safegcd inversion per lane, field30_inv.hip.h).
// pair_kind: what the forward pass found; the denominator it contributes to the shared product.
enum : uint32_t {
    kPairNone = 0,    // no pair in this slot (references in different buckets / end of the segment): denominator 1
    kPairAdd = 1,     // distinct x: den = xb - xa
    kPairDouble = 2,  // same point twice: den = 2 ya, num = 3 xa^2
    kPairCancel = 3,  // opposite points, or both at infinity: the pair contributes nothing
    kPairOnlyA = 4,   // b at infinity
    kPairOnlyB = 5    // a at infinity
};
KZG_HD bool affine30_is_inf(const Affine30& p) { return fq_all_zero(p.x) && fq_all_zero(p.y); }
// full decision with both points at hand (signs applied to y): returns the kind and the denominator
KZG_HD uint32_t pair_classify(const Affine30& a, bool nega, const Affine30& b, bool negb, Fq& den) {
    const bool ainf = affine30_is_inf(a), binf = affine30_is_inf(b);
    den = fq_one();
    if (ainf && binf) return kPairCancel;
    if (ainf) return kPairOnlyB;
    if (binf) return kPairOnlyA;
    const Fq dx = fq_norm(fq_sub_raw(b.x, a.x));
    if (!fq_is_zero(dx)) {
        den = dx;
        return kPairAdd;
    }
    const Fq s = fq_norm(fq_add_raw(fq_cneg(a.y, nega), fq_cneg(b.y, negb)));  // ya + yb: 2 ya or 0
    if (fq_is_zero(s)) return kPairCancel;
    den = s;
    return kPairDouble;
}
// the sum, given 1/den; kind is kPairAdd or kPairDouble.  MUL / SQR: the field product to use (the accumulation
// kernel passes real calls here: its loop body must stay inside the instruction cache)
template <class MulF, class SqrF>
KZG_HD Affine30 pair_sum_with(uint32_t kind, const Affine30& a, bool nega, const Affine30& b, bool negb, const Fq& inv_den,
                              MulF mul, SqrF sqr) {
    const Fq ya = fq_cneg(a.y, nega);
    Fq num;
    if (kind == kPairDouble) {
        const Fq xx = sqr(a.x);
        num = fq_norm(fq_add_raw(fq_add_raw(xx, xx), xx));
    } else {
        num = fq_norm(fq_sub_raw(fq_cneg(b.y, negb), ya));
    }
    KZG_SB30();
    const Fq lambda = mul(num, inv_den);
    KZG_SB30();
    Affine30 r;
    r.x = fq_norm(fq_sub_raw(fq_sub_raw(sqr(lambda), a.x), kind == kPairDouble ? a.x : b.x));
    KZG_SB30();
    r.y = fq_norm(fq_sub_raw(mul(lambda, fq_sub_raw(a.x, r.x)), ya));
    return r;
}
KZG_HD Affine30 pair_sum(uint32_t kind, const Affine30& a, bool nega, const Affine30& b, bool negb, const Fq& inv_den) {
    return pair_sum_with(kind, a, nega, b, negb, inv_den, [](const Fq& x, const Fq& y) { return fq_mul(x, y); },
                         [](const Fq& x) { return fq_sqr(x); });
}

// The general addition as ONE real call with its operands in private memory: the latency-bound finalisation and
// reduction kernels (a dozen dependent additions per lane) then keep one register allocation of ~200 VGPRs for the
// addition and nothing else, instead of several inlined copies whose live ranges the compiler merges into 400+.
#if defined(__HIPCC__)
static __device__ __noinline__ void xyzz30_add_call(XYZZ30* acc, const XYZZ30* b) {
    XYZZ30 a = *acc;
    const XYZZ30 o = *b;
    xyzz30_add(a, o);
    *acc = a;
}
#endif

// ---- the general addition spread over the four lanes of a quad ---------------------------------------------------
// The finalisation and reduction kernels are bound by chains of DEPENDENT additions, and a wave issues one
// instruction every ~4 cycles however many of its lanes work: a lone wave needs ~15 us for the 6650 instructions of one
// general addition (measured: 10.0 us for a mixed one, tools/microbench).  Here the four lanes of a quad hold the SAME
// operands and each computes ONE of the (up to) four independent products of a stage; results travel by DPP quad
// broadcasts (full-rate VALU moves).  Four stages of one product each instead of fourteen products in a row:
//   1  U1 = X1 ZZ2       U2 = X2 ZZ1      S1 = Y1 ZZZ2      S2 = Y2 ZZZ1        -> P = U2 - U1, R = S2 - S1
//   2  PP = P P          RR = R R         ZA = ZZ1 ZZ2      ZB = ZZZ1 ZZZ2
//   3  PPP = P PP        Q = U1 PP        ZZ3 = ZA PP       (lane 3 repeats lane 0)   -> X3 = RR - PPP - 2Q
//   4  ZZZ3 = ZB PPP     T1 = S1 PPP      T2 = R (Q - X3)   (lane 3 repeats lane 0)   -> Y3 = T2 - T1
// Every lane of the quad ends with the same result; all branches (infinity, equal or opposite operands) are taken by
// the whole quad.  Call with all four lanes of every participating quad active.
#if defined(__HIPCC__)
__device__ __forceinline__ Fq fq_quad_select(uint32_t q, const Fq& a0, const Fq& a1, const Fq& a2, const Fq& a3) {
    Fq r;
#pragma unroll
    for (int i = 0; i < kQ; i++) {
        const int32_t lo = (q & 1u) ? a1.d[i] : a0.d[i];
        const int32_t hi = (q & 1u) ? a3.d[i] : a2.d[i];
        r.d[i] = (q & 2u) ? hi : lo;
    }
    return r;
}
template <int SRC>
__device__ __forceinline__ Fq fq_quad_broadcast(const Fq& v) {  // lane SRC of every quad -> all four lanes
    Fq r;
#pragma unroll
    for (int i = 0; i < kQ; i++) {
        r.d[i] = __builtin_amdgcn_mov_dpp(v.d[i], SRC * 0x55, 0xf, 0xf, true);  // quad_perm:[SRC x4]
        // The result stays a plain v_mov_b32_dpp: when LLVM's DPP combiner (ROCm 7.2) folds the move into the
        // subtraction that consumes it (v_subrev_u32_dpp) the lanes of a quad end up with different differences
        // (measured on gfx950: tools/microbench quad; correct with -mllvm -amdgpu-dpp-combine=0).
        asm volatile("" : "+v"(r.d[i]));
    }
    return r;
}
// (A/B switch KZG_QUAD_MUL_CALLS: the four products as calls of ONE copy of the multiplier per kernel instead of four
// inlined ones -- a tenth of the code to fetch on first execution, but 2-10 % more latency at degree 100 ... 2^14.)
#ifdef KZG_QUAD_MUL_CALLS
static __device__ __noinline__ Fq fq_mul_quad_call(Fq a, Fq b) { return fq_mul(a, b); }
#define KZG_QUAD_MUL(a, b) fq_mul_quad_call(a, b)
#else
#define KZG_QUAD_MUL(a, b) fq_mul(a, b)
#endif
// the four stages; `live` = this quad's operands are two finite points (only then may it take the exceptional branch:
// lanes that merely keep the wave busy carry zeros or stale values, whose P may well be zero)
__device__ __forceinline__ void xyzz30_add_quad_core(XYZZ30& acc, const XYZZ30& b, uint32_t q /* lane & 3 */, bool live) {
    // stage 1
    Fq t = KZG_QUAD_MUL(fq_quad_select(q, acc.X, b.X, acc.Y, b.Y), fq_quad_select(q, b.ZZ, acc.ZZ, b.ZZZ, acc.ZZZ));
    const Fq U1 = fq_quad_broadcast<0>(t), U2 = fq_quad_broadcast<1>(t), S1 = fq_quad_broadcast<2>(t), S2 = fq_quad_broadcast<3>(t);
    const Fq P = fq_norm(fq_sub_raw(U2, U1)), R = fq_norm(fq_sub_raw(S2, S1));
    if (live && fq_is_zero(P)) {
        if (fq_is_zero(R)) xyzz30_dbl_inplace(acc);  // every lane doubles its copy
        else acc = xyzz30_inf();
        return;
    }
    // stage 2
    t = KZG_QUAD_MUL(fq_quad_select(q, P, R, acc.ZZ, acc.ZZZ), fq_quad_select(q, P, R, b.ZZ, b.ZZZ));
    const Fq PP = fq_quad_broadcast<0>(t), RR = fq_quad_broadcast<1>(t), ZA = fq_quad_broadcast<2>(t), ZB = fq_quad_broadcast<3>(t);
    // stage 3
    t = KZG_QUAD_MUL(fq_quad_select(q, P, U1, ZA, P), PP);
    const Fq PPP = fq_quad_broadcast<0>(t), Q = fq_quad_broadcast<1>(t);
    acc.ZZ = fq_quad_broadcast<2>(t);
    const Fq X3 = fq_norm_wide(fq_sub_raw(fq_sub_raw(RR, PPP), fq_add_raw(Q, Q)));
    // stage 4
    const Fq QX = fq_norm(fq_sub_raw(Q, X3));
    t = KZG_QUAD_MUL(fq_quad_select(q, ZB, S1, R, ZB), fq_quad_select(q, PPP, PPP, QX, PPP));
    acc.ZZZ = fq_quad_broadcast<0>(t);
    acc.Y = fq_norm(fq_sub_raw(fq_quad_broadcast<2>(t), fq_quad_broadcast<1>(t)));
    acc.X = X3;
}
__device__ __forceinline__ void xyzz30_add_quad(XYZZ30& acc, const XYZZ30& b, uint32_t q /* lane & 3 */) {
    if (xyzz30_is_inf(b)) return;
    if (xyzz30_is_inf(acc)) {
        acc = b;
        return;
    }
    xyzz30_add_quad_core(acc, b, q, true);
}
// The same for a wave in which only some quads have two finite operands.  Measured on MI355X (tools/microbench quad,
// "tree_levels_us"): when the four waves of a workgroup run the addition with at most two quads (8 lanes) active
// each, it takes 17-19 us instead of 5.5-6; with 16 or more active lanes per wave, or with a single wave running, it
// does not.  So the quads that have nothing to add run the SAME instructions on whatever they hold (zeros for infinity;
// they never take the exceptional branch) and keep their own value: every lane of the wave stays active.
// A wave in which no quad has anything to add skips the arithmetic altogether.
__device__ __forceinline__ void xyzz30_add_quad_dense(XYZZ30& acc, const XYZZ30& b, uint32_t q /* lane & 3 */) {
    const bool binf = xyzz30_is_inf(b), ainf = xyzz30_is_inf(acc);
    const bool live = !binf && !ainf;
    if (__builtin_amdgcn_ballot_w64(live) == 0) {
        if (ainf) acc = b;
        return;
    }
    XYZZ30 sum = acc;
    xyzz30_add_quad_core(sum, b, q, live);  // idle quads: same instructions on whatever they hold (zeros at infinity)
    Fq* fo[4] = {&acc.X, &acc.Y, &acc.ZZ, &acc.ZZZ};
    const Fq* fs[4] = {&sum.X, &sum.Y, &sum.ZZ, &sum.ZZZ};
    const Fq* fb[4] = {&b.X, &b.Y, &b.ZZ, &b.ZZZ};
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int i = 0; i < kQ; i++) fo[c]->d[i] = live ? fs[c]->d[i] : (ainf ? fb[c]->d[i] : fo[c]->d[i]);
}
__device__ __forceinline__ Fq load_fq16(const uint4* __restrict__ p) {
    const uint4 a = p[0], b = p[1], c = p[2];
    const uint32_t d = reinterpret_cast<const uint32_t*>(p)[12];
    Fq r;
    r.d[0] = (int32_t)a.x; r.d[1] = (int32_t)a.y; r.d[2] = (int32_t)a.z; r.d[3] = (int32_t)a.w;
    r.d[4] = (int32_t)b.x; r.d[5] = (int32_t)b.y; r.d[6] = (int32_t)b.z; r.d[7] = (int32_t)b.w;
    r.d[8] = (int32_t)c.x; r.d[9] = (int32_t)c.y; r.d[10] = (int32_t)c.z; r.d[11] = (int32_t)c.w;
    r.d[12] = (int32_t)d;
    return r;
}
__device__ __forceinline__ void store_fq16(uint4* __restrict__ p, const Fq& a) {
    p[0] = make_uint4((uint32_t)a.d[0], (uint32_t)a.d[1], (uint32_t)a.d[2], (uint32_t)a.d[3]);
    p[1] = make_uint4((uint32_t)a.d[4], (uint32_t)a.d[5], (uint32_t)a.d[6], (uint32_t)a.d[7]);
    p[2] = make_uint4((uint32_t)a.d[8], (uint32_t)a.d[9], (uint32_t)a.d[10], (uint32_t)a.d[11]);
    p[3] = make_uint4((uint32_t)a.d[12], 0u, 0u, 0u);
}
__device__ __forceinline__ XYZZ30 load_xyzz30(const uint4* __restrict__ rec) {
    XYZZ30 a;
    a.X = load_fq16(rec);
    a.Y = load_fq16(rec + 4);
    a.ZZ = load_fq16(rec + 8);
    a.ZZZ = load_fq16(rec + 12);
    return a;
}
__device__ __forceinline__ void store_xyzz30(uint4* __restrict__ rec, const XYZZ30& a) {
    store_fq16(rec, a.X);
    store_fq16(rec + 4, a.Y);
    store_fq16(rec + 8, a.ZZ);
    store_fq16(rec + 12, a.ZZZ);
}
#endif

}  // namespace kzg
