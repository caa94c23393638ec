// host_pairing.hpp -- BLS12-381 pairing check on the host, part of the PRODUCT library (not the oracle).
//
// Restates Evaluation::verify_proof (reference src/polynomial.rs:276-294), which the reference delegates to
// blst's Miller loop + final exponentiation (src/curves.rs:355-371):
//        e(proof, [s]G2 - [z]G2) == e(commitment - [y]G1, G2).
// two pairings per proof, constant cost, host side.  Written for obviousness, not speed
// (~35 ms per check; blst needs ~1.5 ms): Fp12 = Fp[w]/(w^12 - 2w^6 + 2) with schoolbook products, G2 arithmetic in
// affine Fp2 coordinates on the twist y^2 = x^3 + 4(u+1), Miller loop over |x| = 0xd201000000010000 with affine line
// functions, ONE shared generic exponentiation by (p^12 - 1)/r.  The check is rearranged so that every scalar
// multiplication happens in G1:
//        e(proof, [s]G2) * e(-([z]proof + commitment - [y]G1), G2) == 1,
// which accepts exactly the same (commitment, proof, z, y) as the reference's equation (bilinearity).
// Any non-degenerate bilinear pairing decides the equation, so the sign of the BLS parameter and scalings of the
// line functions by elements of proper subfields (killed by the final exponentiation) do not matter.
#pragma once
#include "host_field.hpp"

namespace kzg_host {

static const Fp kFpR2 = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                          0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};
inline Fp fp_zero() {
    Fp z;
    std::memset(&z, 0, sizeof z);
    return z;
}
inline Fp fp_from_hex(const char* be96) {  // 96 hex digits, big-endian, canonical -> Montgomery
    Fp x = fp_zero();
    for (int i = 0; i < 96; ++i) {
        char ch = be96[i];
        uint64_t v = ch <= '9' ? ch - '0' : (ch | 0x20) - 'a' + 10;
        int bit = 4 * (95 - i);
        x.l[bit >> 6] |= v << (bit & 63);
    }
    return x * kFpR2;
}

// ---- Fp2 = Fp[u] / (u^2 + 1) ------------------------------------------------------------------
struct Fp2 {
    Fp a, b;
    bool is_zero() const { return a.is_zero() && b.is_zero(); }
};
inline Fp2 operator+(const Fp2& x, const Fp2& y) { return {x.a + y.a, x.b + y.b}; }
inline Fp2 operator-(const Fp2& x, const Fp2& y) { return {x.a - y.a, x.b - y.b}; }
inline Fp2 operator*(const Fp2& x, const Fp2& y) { return {x.a * y.a - x.b * y.b, x.a * y.b + x.b * y.a}; }
inline Fp2 fp2_scale(const Fp2& x, const Fp& k) { return {x.a * k, x.b * k}; }
inline Fp2 fp2_neg(const Fp2& x) { return {neg(x.a), neg(x.b)}; }
inline Fp2 fp2_inv(const Fp2& x) {
    Fp d = inv(x.a * x.a + x.b * x.b);
    return {x.a * d, neg(x.b) * d};
}
inline bool operator==(const Fp2& x, const Fp2& y) { return x.a == y.a && x.b == y.b; }

struct G2Affine {  // on the twist; inf flagged
    Fp2 x, y;
    bool inf;
};
inline G2Affine g2_generator() {
    G2Affine g;
    g.x = {fp_from_hex("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"),
           fp_from_hex("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e")};
    g.y = {fp_from_hex("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"),
           fp_from_hex("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be")};
    g.inf = false;
    return g;
}
inline bool g2_on_curve(const G2Affine& q) {  // y^2 = x^3 + 4(u + 1)
    if (q.inf) return true;
    Fp four = kOne + kOne;
    four = four + four;
    Fp2 b2 = {four, four};
    return q.y * q.y == q.x * q.x * q.x + b2;
}
// blst_p2 (Jacobian, Montgomery, 3 x Fp2 = 36 u64; z == 0 <=> infinity) -> affine
inline G2Affine g2_from_p2(const uint64_t* w) {
    Fp2 X, Y, Z;
    std::memcpy(&X, w, 96);
    std::memcpy(&Y, w + 12, 96);
    std::memcpy(&Z, w + 24, 96);
    G2Affine q;
    q.inf = Z.is_zero();
    if (q.inf) {
        q.x = q.y = {fp_zero(), fp_zero()};
        return q;
    }
    Fp2 zi = fp2_inv(Z), zi2 = zi * zi;
    q.x = X * zi2;
    q.y = Y * zi2 * zi;
    return q;
}

// ---- Fp12 = Fp[w] / (w^12 - 2 w^6 + 2);  Fp2 sits inside through u = w^6 - 1 ---------------------
struct F12 {
    Fp c[12];
};
inline F12 f12_zero() {
    F12 r;
    std::memset(&r, 0, sizeof r);
    return r;
}
inline F12 f12_one() {
    F12 r = f12_zero();
    r.c[0] = kOne;
    return r;
}
inline bool f12_is_one(const F12& x) {
    if (!(x.c[0] == kOne)) return false;
    for (int i = 1; i < 12; ++i)
        if (!x.c[i].is_zero()) return false;
    return true;
}
inline F12 operator*(const F12& x, const F12& y) {
    Fp t[23];
    for (auto& v : t) v = fp_zero();
    for (int i = 0; i < 12; ++i) {
        if (x.c[i].is_zero()) continue;
        for (int j = 0; j < 12; ++j) {
            if (y.c[j].is_zero()) continue;
            t[i + j] = t[i + j] + x.c[i] * y.c[j];
        }
    }
    for (int k = 22; k >= 12; --k) {  // w^12 = 2 w^6 - 2
        Fp v2 = t[k] + t[k];
        t[k - 6] = t[k - 6] + v2;
        t[k - 12] = t[k - 12] - v2;
    }
    F12 r;
    for (int i = 0; i < 12; ++i) r.c[i] = t[i];
    return r;
}

inline F12 f12_sqr(const F12& x) {  // cross products once: 78 instead of 144 field products
    Fp t[23];
    for (auto& v : t) v = fp_zero();
    for (int i = 0; i < 12; ++i) {
        if (x.c[i].is_zero()) continue;
        t[2 * i] = t[2 * i] + x.c[i] * x.c[i];
        for (int j = i + 1; j < 12; ++j) {
            if (x.c[j].is_zero()) continue;
            Fp p = x.c[i] * x.c[j];
            t[i + j] = t[i + j] + p + p;
        }
    }
    for (int k = 22; k >= 12; --k) {
        Fp v2 = t[k] + t[k];
        t[k - 6] = t[k - 6] + v2;
        t[k - 12] = t[k - 12] - v2;
    }
    F12 r;
    for (int i = 0; i < 12; ++i) r.c[i] = t[i];
    return r;
}

// Line through T1, T2 (T1 == T2: tangent) on the twist with slope lambda, evaluated at the G1 point (xp, yp) and
// scaled by w^3 (an element of Fp4):  (y1 - lambda x1) + (lambda xp) w^2 - yp w^3,  Fp2 embedded as (a - b) + b w^6.
inline F12 line_value(const Fp2& lambda, const Fp2& x1, const Fp2& y1, const Fp& xp, const Fp& yp) {
    F12 l = f12_zero();
    Fp2 c0 = y1 - lambda * x1;
    Fp2 c2 = fp2_scale(lambda, xp);
    l.c[0] = c0.a - c0.b;
    l.c[6] = c0.b;
    l.c[2] = c2.a - c2.b;
    l.c[8] = c2.b;
    l.c[3] = neg(yp);
    return l;
}

// f_{|x|, Q}(P) without the final exponentiation; 1 when either argument is infinity.  ok = false if the loop
// meets a vertical line (cannot happen for points of order r).
inline F12 miller_loop(const G2Affine& q, const P1& p_jac, bool& ok) {
    ok = true;
    if (q.inf || p_jac.is_inf()) return f12_one();
    P1 pa = p1_normalize(p_jac);
    const Fp xp = pa.x, yp = pa.y;
    const uint64_t ate = 0xd201000000010000ULL;
    Fp2 tx = q.x, ty = q.y;
    F12 f = f12_one();
    Fp three = kOne + kOne + kOne;
    for (int i = 62; i >= 0; --i) {  // bit 63 is the leading one
        if (ty.is_zero()) { ok = false; return f12_one(); }
        Fp2 lambda = fp2_scale(tx * tx, three) * fp2_inv(ty + ty);
        f = f12_sqr(f) * line_value(lambda, tx, ty, xp, yp);
        Fp2 nx = lambda * lambda - tx - tx;
        ty = lambda * (tx - nx) - ty;
        tx = nx;
        if ((ate >> i) & 1) {
            if (tx == q.x) { ok = false; return f12_one(); }
            Fp2 lam = (q.y - ty) * fp2_inv(q.x - tx);
            f = f * line_value(lam, tx, ty, xp, yp);
            Fp2 ax = lam * lam - tx - q.x;
            ty = lam * (tx - ax) - ty;
            tx = ax;
        }
    }
    return f;
}

inline F12 f12_final_exp(const F12& x) {  // x^((p^12 - 1) / r), 4314-bit exponent, 4-bit fixed windows
    static const char* kExp =
        "2ee1db5dcc825b7e1bda9c0496a1c0a89ee0193d4977b3f7d4507d07363baa13f8d14a917848517badc3a43d1073776a"
        "b353f2c30698e8cc7deada9c0aadff5e9cfee9a074e43b9a660835cc872ee83ff3a0f0f1c0ad0d6106feaf4e347aa68a"
        "d49466fa927e7bb9375331807a0dce2630d9aa4b113f414386b0e8819328148978e2b0dd39099b86e1ab656d2670d93e"
        "4d7acdd350da5359bc73ab61a0c5bf24c374693c49f570bcd2b01f3077ffb10bf24dde41064837f27611212596bc293c"
        "8d4c01f25118790f4684d0b9c40a68eb74bb22a40ee7169cdc1041296532fef459f12438dfc8e2886ef965e61a474c5c"
        "85b0129127a1b5ad0463434724538411d1676a53b5a62eb34c05739334f46c02c3f0bd0c55d3109cd15948d0a1fad200"
        "44ce6ad4c6bec3ec03ef19592004cedd556952c6d8823b19dadd7c2498345c6e5308f1c511291097db60b1749bf9b71a"
        "9f9e0100418a3ef0bc627751bbd81367066bca6a4c1b6dcfc5cceb73fc56947a403577dfa9e13c24ea820b09c1d9f7c3"
        "1759c3635de3f7a3639991708e88adce88177456c49637fd7961be1a4c7e79fb02faa732e2f3ec2bea83d19628331349"
        "2caa9d4aff1c910e9622d2a73f62537f2701aaef6539314043f7bbce5b78c7869aeb2181a67e49eeed2161daf3f881bd"
        "88592d767f67c4717489119226c2f011d4cab803e9d71650a6f80698e2f8491d12191a04406fbc8fbd5f48925f98630e"
        "68bfb24c0bcb9b55df57510";
    F12 pw[16];
    pw[0] = f12_one();
    for (int i = 1; i < 16; ++i) pw[i] = pw[i - 1] * x;
    F12 acc = f12_one();
    for (const char* c = kExp; *c; ++c) {
        int v = *c <= '9' ? *c - '0' : *c - 'a' + 10;
        acc = f12_sqr(f12_sqr(f12_sqr(f12_sqr(acc))));
        if (v) acc = acc * pw[v];
    }
    return acc;
}

// k * P for a canonical little-endian 256-bit scalar (4 x u64)
inline P1 p1_mul(const P1& p, const uint64_t k[4]) {
    P1 acc = p1_inf();
    for (int i = 255; i >= 0; --i) {
        acc = p1_double(acc);
        if ((k[i >> 6] >> (i & 63)) & 1) acc = p1_add(acc, p);
    }
    return acc;
}
inline P1 p1_neg(const P1& p) {
    P1 r = p;
    r.y = neg(p.y);
    return r;
}
inline P1 p1_generator() {
    P1 g;
    g.x = fp_from_hex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb");
    g.y = fp_from_hex("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1");
    g.z = kOne;
    return g;
}

// canonical integer (4 x u64 LE) of a blst_fr in Montgomery form (R = 2^256): one Montgomery reduction
inline void fr_from_mont(const uint64_t a[4], uint64_t out[4]) {
    static const uint64_t r[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    const uint64_t n0 = 0xfffffffeffffffffULL;
    uint64_t t[9] = {a[0], a[1], a[2], a[3], 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        uint64_t m = t[i] * n0;
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)m * r[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        for (int k = i + 4; c != 0 && k < 9; ++k) {
            c += t[k];
            t[k] = (uint64_t)c;
            c >>= 64;
        }
    }
    uint64_t o[4] = {t[4], t[5], t[6], t[7]};
    // conditional subtraction of r
    uint64_t d[4];
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 v = (u128)o[i] - r[i] - br;
        d[i] = (uint64_t)v;
        br = (uint64_t)(v >> 64) & 1;
    }
    bool ge = t[8] != 0 || br == 0;
    for (int i = 0; i < 4; ++i) out[i] = ge ? d[i] : o[i];
}

// Evaluation::verify_proof.  commitment / proof: blst_p1; z, y: blst_fr (Montgomery); s_g2: blst_p2 = [s]G2.
// Returns 1 accepted, 0 rejected, -1 malformed input (G2 point not on the twist).
inline int verify_proof(const uint64_t* commitment, const uint64_t* proof, const uint64_t z_mont[4], const uint64_t y_mont[4],
                        const uint64_t* s_g2) {
    P1 C, Pi;
    std::memcpy(&C, commitment, sizeof C);
    std::memcpy(&Pi, proof, sizeof Pi);
    uint64_t z[4], y[4];
    fr_from_mont(z_mont, z);
    fr_from_mont(y_mont, y);
    G2Affine sg2 = g2_from_p2(s_g2);
    if (!g2_on_curve(sg2)) return -1;
    // rhs = [z]proof + commitment - [y]G1
    P1 rhs = p1_add(p1_add(p1_mul(Pi, z), C), p1_neg(p1_mul(p1_generator(), y)));
    bool ok1 = true, ok2 = true;
    F12 f = miller_loop(sg2, Pi, ok1) * miller_loop(g2_generator(), p1_neg(rhs), ok2);
    if (!ok1 || !ok2) return 0;
    return f12_is_one(f12_final_exp(f)) ? 1 : 0;
}

}  // namespace kzg_host
