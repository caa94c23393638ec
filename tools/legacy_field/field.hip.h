// field.hip.h -- BLS12-381 base field Fp (381 bit) and scalar field Fr (255 bit) for gfx950.
//
// Representation: little-endian 32-bit limbs (Fp: 12, Fr: 8), Montgomery form with R = 2^384
// (Fp) / 2^256 (Fr).  Twelve u32 limbs are byte-identical to blst's 6 x u64 `blst_fp`
// (reference src/curves.rs:10-17 wraps blst_p1 = 3 x blst_fp), eight u32 limbs to `blst_fr`
// (reference src/scalar.rs:7-8), so buffers cross the C-ABI without conversion.
//
// CDNA4 has no carry-in on its 32x32->64 multiply-add (v_mad_u64_u32 D = S0*S1 + S2), and that
// instruction costs about two simple 32-bit VALU instructions (measured, tools/microbench mix).
// The multiplier is therefore written as
// word-serial Montgomery (one row of a*b_i and one row of m*p per step), each row produced by a
// chain of v_mad_u64_u32 whose high word feeds the next one (no separate carry instruction), and
// folded into the accumulator by one v_addc_co_u32 chain: 2*n^2 multiply-adds + ~2*n^2 adds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kzg {

typedef uint32_t u32;
typedef uint64_t u64;

#define KZG_DEV __device__ __forceinline__

// ---- carry primitives ---------------------------------------------------------------------
KZG_DEV u32 addc(u32 a, u32 b, u32& carry) {  // a + b + carry, carry in {0,1}
    u32 co;
    u32 r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
}
KZG_DEV u32 subb(u32 a, u32 b, u32& borrow) {  // a - b - borrow
    u32 bo;
    u32 r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
}
KZG_DEV u64 mad64(u32 a, u32 b, u64 c) { return (u64)a * b + c; }  // -> v_mad_u64_u32

// ---- field parameter packs ----------------------------------------------------------------
struct FpParams {
    static constexpr int N = 12;
    static constexpr u32 N0 = 0xfffcfffdu;
    KZG_DEV static u32 mod(int i) {
        constexpr u32 P[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                               0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return P[i];
    }
    KZG_DEV static u32 one(int i) {  // R mod p
        constexpr u32 V[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                               0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return V[i];
    }
    KZG_DEV static u32 r2(int i) {  // R^2 mod p
        constexpr u32 V[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                               0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return V[i];
    }
};
struct FrParams {
    static constexpr int N = 8;
    static constexpr u32 N0 = 0xffffffffu;
    KZG_DEV static u32 mod(int i) {
        constexpr u32 P[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                              0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return P[i];
    }
    KZG_DEV static u32 one(int i) {
        constexpr u32 V[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                              0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return V[i];
    }
    KZG_DEV static u32 r2(int i) {
        constexpr u32 V[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                              0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return V[i];
    }
};

// ---- generic Montgomery field element -----------------------------------------------------
template <class F>
struct Fe {
    static constexpr int N = F::N;
    u32 l[F::N];

    KZG_DEV static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    KZG_DEV static Fe one() {
        Fe r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = F::one(i);
        return r;
    }
    KZG_DEV bool is_zero() const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i];
        return o == 0;
    }
    KZG_DEV bool operator==(const Fe& b) const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < N; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
};

// r = a - mod if a >= mod (a < 2*mod, `top` = carry word above limb N-1)
template <class F>
KZG_DEV void cond_sub_mod(Fe<F>& a, u32 top) {
    constexpr int N = F::N;
    u32 d[N];
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) d[i] = subb(a.l[i], F::mod(i), br);
    // a >= mod  <=>  (top:a) - mod did not borrow out of the top word
    bool ge = top >= br;  // top in {0,1}: top - br >= 0
#pragma unroll
    for (int i = 0; i < N; i++) a.l[i] = ge ? d[i] : a.l[i];
}

template <class F>
KZG_DEV Fe<F> fe_add(const Fe<F>& a, const Fe<F>& b) {
    constexpr int N = F::N;
    Fe<F> r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = addc(a.l[i], b.l[i], c);
    cond_sub_mod(r, c);
    return r;
}

template <class F>
KZG_DEV Fe<F> fe_sub(const Fe<F>& a, const Fe<F>& b) {
    constexpr int N = F::N;
    Fe<F> r;
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = subb(a.l[i], b.l[i], br);
    u32 mask = 0u - br;  // all ones when a < b
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = addc(r.l[i], F::mod(i) & mask, c);
    return r;
}

template <class F>
KZG_DEV Fe<F> fe_neg(const Fe<F>& a) {
    constexpr int N = F::N;
    Fe<F> r;
    u32 br = 0;
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= a.l[i];
    u32 mask = nz ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = subb(F::mod(i) & mask, a.l[i], br);
    return r;
}

template <class F>
KZG_DEV Fe<F> fe_dbl(const Fe<F>& a) {
    return fe_add(a, a);
}

// Montgomery product a*b/R mod m, fully reduced.  Accepts any a < 2^(32N), b < m.
template <class F>
KZG_DEV Fe<F> fe_mul(const Fe<F>& a, const Fe<F>& b) {
    constexpr int N = F::N;
    u32 t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;

#pragma unroll
    for (int i = 0; i < N; i++) {
        // row = a * b_i : N+1 words, produced by a mad chain (high word carried in S2)
        u32 row[N + 1];
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            c = mad64(a.l[j], b.l[i], c);
            row[j] = (u32)c;
            c >>= 32;
        }
        row[N] = (u32)c;
        u32 cy = 0;
#pragma unroll
        for (int j = 0; j <= N; j++) t[j] = addc(t[j], row[j], cy);
        t[N + 1] = cy;

        // m = t0 * n0 ; t = (t + m * mod) >> 32
        u32 m = t[0] * F::N0;
        c = mad64(m, F::mod(0), (u64)t[0]);  // low word becomes zero by construction
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c = mad64(m, F::mod(j), c);
            row[j - 1] = (u32)c;
            c >>= 32;
        }
        row[N - 1] = (u32)c;
        // t[1..N+1] + row[0..N-1] -> t[0..N]
        cy = 0;
#pragma unroll
        for (int j = 0; j < N; j++) t[j] = addc(t[j + 1], row[j], cy);
        t[N] = t[N + 1] + cy;
    }
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    cond_sub_mod(r, t[N]);
    return r;
}

template <class F>
KZG_DEV Fe<F> fe_sqr(const Fe<F>& a) {
    return fe_mul(a, a);
}

// Montgomery -> canonical integer in [0, m): multiply by 1 (N reduction rows only).
template <class F>
KZG_DEV Fe<F> fe_from_mont(const Fe<F>& a) {
    constexpr int N = F::N;
    u32 t[N + 1];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = a.l[i];
    t[N] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u32 m = t[0] * F::N0;
        u64 c = mad64(m, F::mod(0), (u64)t[0]);
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c = mad64(m, F::mod(j), c + t[j]);
            t[j - 1] = (u32)c;
            c >>= 32;
        }
        t[N - 1] = (u32)c;
    }
    Fe<F> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    cond_sub_mod(r, 0u);
    return r;
}

template <class F>
KZG_DEV Fe<F> fe_to_mont(const Fe<F>& a) {
    Fe<F> r2;
#pragma unroll
    for (int i = 0; i < F::N; i++) r2.l[i] = F::r2(i);
    return fe_mul(a, r2);
}

typedef Fe<FpParams> Fp;
typedef Fe<FrParams> Fr;

// ---- lazily reduced Fp: representatives in [0, 2p) ------------------------------------------------
// A Montgomery product of two values below 2p is below (4p^2 + R p)/R < 1.41 p, so the conditional
// subtraction that ends every multiplication can be dropped if additions and subtractions wrap at 2p
// instead of p (2p < 2^382 still fits twelve limbs with room for a sum of two).  Zero then has two
// representatives, 0 and p.  Values are made canonical (fp_canon) only where they leave the device
// arithmetic: SRS read-back and the handful of partial sums handed to the host.
KZG_DEV u32 fp_2p(int i) {
    constexpr u32 V[12] = {0xffff5556u, 0x73fdffffu, 0x62a7ffffu, 0x3d57fffdu, 0xed61ec48u, 0xce61a541u,
                           0xe70a257eu, 0xc8ee9709u, 0x869759aeu, 0x96374f6cu, 0x72ffcd34u, 0x340223d4u};
    return V[i];
}
KZG_DEV Fp fp_canon(Fp a) {  // [0, 2p) -> [0, p)
    cond_sub_mod(a, 0u);
    return a;
}
KZG_DEV Fp fp_add_lz(const Fp& a, const Fp& b) {
    Fp r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = addc(a.l[i], b.l[i], c);  // < 4p < 2^384: no carry out
    u32 d[12];
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) d[i] = subb(r.l[i], fp_2p(i), br);
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = br ? r.l[i] : d[i];
    return r;
}
KZG_DEV Fp fp_sub_lz(const Fp& a, const Fp& b) {
    Fp r;
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = subb(a.l[i], b.l[i], br);
    u32 mask = 0u - br;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = addc(r.l[i], fp_2p(i) & mask, c);
    return r;
}
KZG_DEV Fp fp_neg_lz(const Fp& a) {  // 2p - a, and 0 stays 0
    Fp r;
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) nz |= a.l[i];
    u32 mask = nz ? 0xffffffffu : 0u;
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = subb(fp_2p(i) & mask, a.l[i], br);
    return r;
}
KZG_DEV bool fp_is_zero_lz(const Fp& a) {  // a == 0 or a == p
    u32 z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        z |= a.l[i];
        e |= a.l[i] ^ FpParams::mod(i);
    }
    return z == 0 || e == 0;
}

// ---- rolled (loop, scratch-resident) Fp arithmetic for the exceptional branches -------------
// The group law's rare branches (doubling when a bucket receives the point it already holds)
// must not set the register budget of the hot loop, so they run on small non-unrolled routines
// that work on arrays in private memory.  Slow by design; bit-identical results.
static __device__ const u32 KZG_FP_MOD[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                       0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};

static __device__ __noinline__ void fp_rolled_condsub(u32* t, u32 top) {
    u32 d[12];
    u32 br = 0;
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        u64 v = (u64)t[i] - KZG_FP_MOD[i] - br;
        d[i] = (u32)v;
        br = (u32)(v >> 32) & 1u;
    }
    if (top >= br) {
#pragma unroll 1
        for (int i = 0; i < 12; i++) t[i] = d[i];
    }
}
static __device__ __noinline__ void fp_rolled_mul(u32* r, const u32* a, const u32* b) {
    u32 t[14];
#pragma unroll 1
    for (int i = 0; i < 14; i++) t[i] = 0;
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        u64 c = 0;
        u32 bi = b[i];
#pragma unroll 1
        for (int j = 0; j < 12; j++) {
            c += (u64)a[j] * bi + t[j];
            t[j] = (u32)c;
            c >>= 32;
        }
        c += t[12];
        t[12] = (u32)c;
        t[13] = (u32)(c >> 32);
        u32 m = t[0] * FpParams::N0;
        c = ((u64)m * KZG_FP_MOD[0] + t[0]) >> 32;
#pragma unroll 1
        for (int j = 1; j < 12; j++) {
            c += (u64)m * KZG_FP_MOD[j] + t[j];
            t[j - 1] = (u32)c;
            c >>= 32;
        }
        c += t[12];
        t[11] = (u32)c;
        t[12] = t[13] + (u32)(c >> 32);
    }
    fp_rolled_condsub(t, t[12]);
#pragma unroll 1
    for (int i = 0; i < 12; i++) r[i] = t[i];
}
static __device__ __noinline__ void fp_rolled_add(u32* r, const u32* a, const u32* b) {
    u32 t[12];
    u64 c = 0;
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        c += (u64)a[i] + b[i];
        t[i] = (u32)c;
        c >>= 32;
    }
    fp_rolled_condsub(t, (u32)c);
#pragma unroll 1
    for (int i = 0; i < 12; i++) r[i] = t[i];
}
static __device__ __noinline__ void fp_rolled_sub(u32* r, const u32* a, const u32* b) {
    u32 t[12];
    u32 br = 0;
#pragma unroll 1
    for (int i = 0; i < 12; i++) {
        u64 v = (u64)a[i] - b[i] - br;
        t[i] = (u32)v;
        br = (u32)(v >> 32) & 1u;
    }
    if (br) {
        u64 c = 0;
#pragma unroll 1
        for (int i = 0; i < 12; i++) {
            c += (u64)t[i] + KZG_FP_MOD[i];
            t[i] = (u32)c;
            c >>= 32;
        }
    }
#pragma unroll 1
    for (int i = 0; i < 12; i++) r[i] = t[i];
}

// a^(p-2): Fermat inversion (only used in the per-thread batch-inversion prologue/epilogue).
KZG_DEV Fp fp_inv(const Fp& a) {
    // p - 2, little-endian 32-bit words
    constexpr u32 E[12] = {0xffffaaa9u, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                           0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    Fp acc = Fp::one();
    // left-to-right binary; top word has 29 significant bits (0x1a0111ea)
    for (int w = 11; w >= 0; w--) {
        u32 e = E[w];
        for (int b = 31; b >= 0; b--) {
            acc = fe_sqr(acc);
            if ((e >> b) & 1) acc = fe_mul(acc, a);
        }
    }
    return acc;
}

}  // namespace kzg
