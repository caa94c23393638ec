// host_pairing.hpp -- BLS12-381 pairing check on the host, part of the PRODUCT library (not the oracle).
//
// Restates Evaluation::verify_proof (reference src/polynomial.rs:276-294), which the reference delegates to
// blst's Miller loop + final exponentiation (src/curves.rs:355-371):
//        e(proof, [s]G2 - [z]G2) == e(commitment - [y]G1, G2).
// two pairings per proof, constant cost, host side.  Tower Fp2 -> Fp6 -> Fp12 (w^6 = 1 + u),
// ONE Miller loop over |x| = 0xd201000000010000 for both pairings (shared squaring, division-free projective line
// functions on the twist y^2 = x^3 + 4(u+1), sparse 0-1-4 products), final exponentiation = easy part + five
// exponentiations by |x| with Granger-Scott cyclotomic squarings.  ~25 000 base-field products per check.
// The check is rearranged so that every scalar multiplication happens in G1:
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

// ---- tower Fp2 -> Fp6 = Fp2[v]/(v^3 - xi) -> Fp12 = Fp6[w]/(w^2 - v), xi = 1 + u -----------------------------
// (so w^6 = xi: an element is sum_{i<6} a_i w^i with a_i in Fp2; c0 = (a0, a2, a4), c1 = (a1, a3, a5))
inline Fp2 fp2_mul(const Fp2& x, const Fp2& y) {  // Karatsuba: 3 base-field products
    const Fp t0 = x.a * y.a, t1 = x.b * y.b;
    return {t0 - t1, (x.a + x.b) * (y.a + y.b) - t0 - t1};
}
inline Fp2 fp2_sqr(const Fp2& x) {  // (a + b)(a - b), 2ab
    const Fp ab = x.a * x.b;
    return {(x.a + x.b) * (x.a - x.b), ab + ab};
}
inline Fp2 fp2_mul_xi(const Fp2& x) { return {x.a - x.b, x.a + x.b}; }  // (a + bu)(1 + u)
inline Fp2 fp2_conj(const Fp2& x) { return {x.a, neg(x.b)}; }
inline Fp2 fp2_dbl(const Fp2& x) { return x + x; }
inline Fp2 fp2_zero() { return {fp_zero(), fp_zero()}; }
inline Fp2 fp2_one() { return {kOne, fp_zero()}; }

struct Fp6 {
    Fp2 c0, c1, c2;
};
inline Fp6 operator+(const Fp6& x, const Fp6& y) { return {x.c0 + y.c0, x.c1 + y.c1, x.c2 + y.c2}; }
inline Fp6 operator-(const Fp6& x, const Fp6& y) { return {x.c0 - y.c0, x.c1 - y.c1, x.c2 - y.c2}; }
inline Fp6 fp6_neg(const Fp6& x) { return {fp2_neg(x.c0), fp2_neg(x.c1), fp2_neg(x.c2)}; }
inline Fp6 fp6_mul_v(const Fp6& x) { return {fp2_mul_xi(x.c2), x.c0, x.c1}; }
inline Fp6 fp6_mul(const Fp6& a, const Fp6& b) {  // Karatsuba, 6 Fp2 products
    const Fp2 t0 = fp2_mul(a.c0, b.c0), t1 = fp2_mul(a.c1, b.c1), t2 = fp2_mul(a.c2, b.c2);
    Fp6 r;
    r.c0 = t0 + fp2_mul_xi(fp2_mul(a.c1 + a.c2, b.c1 + b.c2) - t1 - t2);
    r.c1 = fp2_mul(a.c0 + a.c1, b.c0 + b.c1) - t0 - t1 + fp2_mul_xi(t2);
    r.c2 = fp2_mul(a.c0 + a.c2, b.c0 + b.c2) - t0 - t2 + t1;
    return r;
}
inline Fp6 fp6_mul_by_01(const Fp6& a, const Fp2& b0, const Fp2& b1) {  // a * (b0 + b1 v)
    return {fp2_mul(a.c0, b0) + fp2_mul_xi(fp2_mul(a.c2, b1)), fp2_mul(a.c0, b1) + fp2_mul(a.c1, b0),
            fp2_mul(a.c1, b1) + fp2_mul(a.c2, b0)};
}
inline Fp6 fp6_mul_by_1(const Fp6& a, const Fp2& b1) {  // a * (b1 v)
    return {fp2_mul_xi(fp2_mul(a.c2, b1)), fp2_mul(a.c0, b1), fp2_mul(a.c1, b1)};
}
inline Fp6 fp6_inv(const Fp6& a) {
    const Fp2 c0 = fp2_sqr(a.c0) - fp2_mul_xi(fp2_mul(a.c1, a.c2));
    const Fp2 c1 = fp2_mul_xi(fp2_sqr(a.c2)) - fp2_mul(a.c0, a.c1);
    const Fp2 c2 = fp2_sqr(a.c1) - fp2_mul(a.c0, a.c2);
    const Fp2 t = fp2_inv(fp2_mul(a.c0, c0) + fp2_mul_xi(fp2_mul(a.c2, c1) + fp2_mul(a.c1, c2)));
    return {fp2_mul(c0, t), fp2_mul(c1, t), fp2_mul(c2, t)};
}

struct F12 {
    Fp6 c0, c1;
};
inline F12 f12_one() {
    F12 r;
    std::memset(&r, 0, sizeof r);
    r.c0.c0.a = kOne;
    return r;
}
inline bool f12_is_one(const F12& x) {
    const F12 one = f12_one();
    return std::memcmp(&x, &one, sizeof x) == 0;  // base-field values are canonical
}
inline F12 operator*(const F12& a, const F12& b) {  // 3 Fp6 products
    const Fp6 t0 = fp6_mul(a.c0, b.c0), t1 = fp6_mul(a.c1, b.c1);
    return {t0 + fp6_mul_v(t1), fp6_mul(a.c0 + a.c1, b.c0 + b.c1) - t0 - t1};
}
inline F12 f12_sqr(const F12& a) {  // complex squaring, 2 Fp6 products
    const Fp6 ab = fp6_mul(a.c0, a.c1);
    return {fp6_mul(a.c0 + a.c1, a.c0 + fp6_mul_v(a.c1)) - ab - fp6_mul_v(ab), ab + ab};
}
inline F12 f12_conj(const F12& a) { return {a.c0, fp6_neg(a.c1)}; }  // the p^6 Frobenius; the inverse on the cyclotomic subgroup
inline F12 f12_inv(const F12& a) {
    const Fp6 t = fp6_inv(fp6_mul(a.c0, a.c0) - fp6_mul_v(fp6_mul(a.c1, a.c1)));
    return {fp6_mul(a.c0, t), fp6_neg(fp6_mul(a.c1, t))};
}
// a * (l0 + l1 v + l4 v w): the shape of every Miller-loop line (sparse "0-1-4" product, 13 Fp2 products)
inline F12 f12_mul_by_014(const F12& a, const Fp2& l0, const Fp2& l1, const Fp2& l4) {
    const Fp6 t0 = fp6_mul_by_01(a.c0, l0, l1), t1 = fp6_mul_by_1(a.c1, l4);
    return {fp6_mul_v(t1) + t0, fp6_mul_by_01(a.c0 + a.c1, l0, l1 + l4) - t0 - t1};
}
// Granger-Scott squaring: valid for elements of the cyclotomic subgroup (after the easy part of the final
// exponentiation), 9 Fp2 squarings' worth instead of 12 Fp2 products
inline void fp4_sqr(const Fp2& a, const Fp2& b, Fp2& t0, Fp2& t1) {  // (a + b y)^2, y^2 = xi
    const Fp2 ab = fp2_mul(a, b);
    t0 = fp2_mul(a + b, fp2_mul_xi(b) + a) - ab - fp2_mul_xi(ab);
    t1 = ab + ab;
}
inline F12 f12_cyclotomic_sqr(const F12& f) {
    const Fp2 &r0 = f.c0.c0, &r4 = f.c0.c1, &r3 = f.c0.c2, &r2 = f.c1.c0, &r1 = f.c1.c1, &r5 = f.c1.c2;
    Fp2 t0, t1, t2, t3, t4, t5;
    fp4_sqr(r0, r1, t0, t1);
    fp4_sqr(r2, r3, t2, t3);
    fp4_sqr(r4, r5, t4, t5);
    F12 z;
    auto three_minus_two = [](const Fp2& t, const Fp2& r) { const Fp2 d = t - r; return d + d + t; };  // 3t - 2r
    auto three_plus_two = [](const Fp2& t, const Fp2& r) { const Fp2 d = t + r; return d + d + t; };    // 3t + 2r
    z.c0.c0 = three_minus_two(t0, r0);
    z.c1.c1 = three_plus_two(t1, r1);
    z.c1.c0 = three_plus_two(fp2_mul_xi(t5), r2);
    z.c0.c2 = three_minus_two(t4, r3);
    z.c0.c1 = three_minus_two(t2, r4);
    z.c1.c2 = three_plus_two(t3, r5);
    return z;
}
// Frobenius x -> x^p: every Fp2 coefficient is conjugated and the coefficient of w^i picks up gamma^i with
// gamma = w^(p-1) = xi^((p-1)/6) in Fp2 (computed once)
inline const Fp2* frobenius_gammas() {
    static Fp2 g[6];
    static bool ready = false;
    if (!ready) {
        // (p - 1) / 6 from kP: p - 1 is divisible by 6
        uint64_t q[6], rem = 0;
        Fp pm1 = kP;
        pm1.l[0] -= 1;
        for (int i = 5; i >= 0; --i) {
            u128 cur = ((u128)rem << 64) | pm1.l[i];
            q[i] = (uint64_t)(cur / 6);
            rem = (uint64_t)(cur % 6);
        }
        Fp2 base = {kOne, kOne}, acc = fp2_one();  // xi = 1 + u
        for (int i = 0; i < 384; ++i) {
            if ((q[i >> 6] >> (i & 63)) & 1) acc = fp2_mul(acc, base);
            base = fp2_sqr(base);
        }
        g[0] = fp2_one();
        for (int i = 1; i < 6; ++i) g[i] = fp2_mul(g[i - 1], acc);
        ready = true;
    }
    return g;
}
inline F12 f12_frobenius(const F12& a) {
    const Fp2* g = frobenius_gammas();
    F12 r;
    r.c0.c0 = fp2_conj(a.c0.c0);                    // w^0
    r.c1.c0 = fp2_mul(fp2_conj(a.c1.c0), g[1]);     // w^1
    r.c0.c1 = fp2_mul(fp2_conj(a.c0.c1), g[2]);     // w^2
    r.c1.c1 = fp2_mul(fp2_conj(a.c1.c1), g[3]);     // w^3
    r.c0.c2 = fp2_mul(fp2_conj(a.c0.c2), g[4]);     // w^4
    r.c1.c2 = fp2_mul(fp2_conj(a.c1.c2), g[5]);     // w^5
    return r;
}

// ---- Miller loop ---------------------------------------------------------------------------------
// T on the twist in homogeneous projective coordinates (x = X/Z, y = Y/Z), P = (xp, yp) in G1.  With slope
// lambda the line through T (tangent, or chord to Q) evaluated at P and scaled by w^3 is
//        (y_T - lambda x_T) + (lambda xp) w^2 - yp w^3 ;
// multiplied by the slope's denominator (an element of Fp2, which the final exponentiation kills) it has no
// division:  tangent  (2Y^2 Z - 3X^3) + (3 X^2 Z xp) w^2 - (2 Y Z^2 yp) w^3
//            chord    (y_Q mu - theta x_Q) + (theta xp) w^2 - (mu yp) w^3,   theta = y_Q Z - Y, mu = x_Q Z - X.
struct G2Proj {
    Fp2 X, Y, Z;
};
struct LineCoeffs {
    Fp2 l0, l1, l4;  // coefficients of w^0, w^2 (= v), w^3 (= v w)
};
inline LineCoeffs miller_double(G2Proj& t, const Fp& xp, const Fp& yp) {
    const Fp2 XX = fp2_sqr(t.X), YY = fp2_sqr(t.Y), ZZ = fp2_sqr(t.Z);
    const Fp2 X3 = fp2_mul(XX, t.X);
    const Fp2 W = XX + XX + XX;                 // 3X^2
    LineCoeffs l;
    l.l0 = fp2_dbl(fp2_mul(YY, t.Z)) - (X3 + X3 + X3);
    l.l1 = fp2_scale(fp2_mul(W, t.Z), xp);
    l.l4 = fp2_neg(fp2_scale(fp2_dbl(fp2_mul(t.Y, ZZ)), yp));
    // doubling (a = 0, homogeneous): s = 2YZ, B = 2 X Y s, h = W^2 - 2B
    const Fp2 S = fp2_dbl(fp2_mul(t.Y, t.Z));
    const Fp2 R = fp2_mul(t.Y, S);
    const Fp2 B = fp2_dbl(fp2_mul(t.X, R));
    const Fp2 H = fp2_sqr(W) - fp2_dbl(B);
    const Fp2 RR = fp2_sqr(R);
    const Fp2 SS = fp2_sqr(S);
    t.Y = fp2_mul(W, B - H) - fp2_dbl(RR);
    t.X = fp2_mul(H, S);
    t.Z = fp2_mul(S, SS);
    return l;
}
inline LineCoeffs miller_add(G2Proj& t, const G2Affine& q, const Fp& xp, const Fp& yp) {
    const Fp2 theta = fp2_mul(q.y, t.Z) - t.Y;  // slope numerator
    const Fp2 mu = fp2_mul(q.x, t.Z) - t.X;     // slope denominator
    LineCoeffs l;
    l.l0 = fp2_mul(q.y, mu) - fp2_mul(theta, q.x);
    l.l1 = fp2_scale(theta, xp);
    l.l4 = fp2_neg(fp2_scale(mu, yp));
    // mixed addition (madd-1998-cmo): u = theta, v = mu
    const Fp2 vv = fp2_sqr(mu), vvv = fp2_mul(vv, mu);
    const Fp2 Rr = fp2_mul(vv, t.X);
    const Fp2 A = fp2_mul(fp2_sqr(theta), t.Z) - vvv - fp2_dbl(Rr);
    t.Y = fp2_mul(theta, Rr - A) - fp2_mul(vvv, t.Y);
    t.X = fp2_mul(mu, A);
    t.Z = fp2_mul(vvv, t.Z);
    return l;
}

// prod_k f_{|x|, Q_k}(P_k) without the final exponentiation, one shared squaring per loop step; pairs with a
// point at infinity contribute 1.  ok = false if a chord degenerates (cannot happen for points of order r).
struct MillerPair {
    G2Affine q;
    Fp xp, yp;
    G2Proj t;
    bool live;
};
inline F12 multi_miller_loop(const G2Affine* qs, const P1* ps, int count, bool& ok) {
    ok = true;
    MillerPair pr[4];
    for (int k = 0; k < count; ++k) {
        pr[k].live = !(qs[k].inf || ps[k].is_inf());
        if (!pr[k].live) continue;
        const P1 pa = p1_normalize(ps[k]);
        pr[k].q = qs[k];
        pr[k].xp = pa.x;
        pr[k].yp = pa.y;
        pr[k].t = {qs[k].x, qs[k].y, fp2_one()};
    }
    const uint64_t ate = 0xd201000000010000ULL;
    F12 f = f12_one();
    for (int i = 62; i >= 0; --i) {  // bit 63 is the leading one
        f = f12_sqr(f);
        for (int k = 0; k < count; ++k) {
            if (!pr[k].live) continue;
            if (pr[k].t.Y.is_zero() || pr[k].t.Z.is_zero()) { ok = false; return f12_one(); }
            const LineCoeffs l = miller_double(pr[k].t, pr[k].xp, pr[k].yp);
            f = f12_mul_by_014(f, l.l0, l.l1, l.l4);
        }
        if ((ate >> i) & 1) {
            for (int k = 0; k < count; ++k) {
                if (!pr[k].live) continue;
                const LineCoeffs l = miller_add(pr[k].t, pr[k].q, pr[k].xp, pr[k].yp);
                if (pr[k].t.Z.is_zero()) { ok = false; return f12_one(); }
                f = f12_mul_by_014(f, l.l0, l.l1, l.l4);
            }
        }
    }
    return f;
}

// x -> x^|z| for the BLS parameter |z| = 0xd201000000010000 on the cyclotomic subgroup
inline F12 f12_cyclotomic_exp_z(const F12& a) {
    const uint64_t z = 0xd201000000010000ULL;
    F12 acc = a;
    for (int i = 62; i >= 0; --i) {
        acc = f12_cyclotomic_sqr(acc);
        if ((z >> i) & 1) acc = acc * a;
    }
    return acc;
}
// x^(3 (p^12 - 1) / r).  The factor 3 does not change whether the result is 1 (the result lies in the subgroup of
// order r, and gcd(3, r) = 1).  Easy part (p^6 - 1)(p^2 + 1); hard part by
//        3 (p^4 - p^2 + 1) / r = (z - 1)^2 (z + p) (z^2 + p^2 - 1) + 3,   z = -|z|,
// i.e. five exponentiations by |z| (63 cyclotomic squarings each), conjugations (= inverses) and Frobenius maps.
inline F12 f12_final_exp(const F12& x) {
    F12 a = f12_conj(x) * f12_inv(x);            // x^(p^6 - 1)
    a = f12_frobenius(f12_frobenius(a)) * a;     // ^(p^2 + 1): now in the cyclotomic subgroup
    auto exp_z = [](const F12& v) { return f12_conj(f12_cyclotomic_exp_z(v)); };  // v^z, z negative
    const F12 A = exp_z(a) * f12_conj(a);                                   // a^(z - 1)
    const F12 B = exp_z(A) * f12_conj(A);                                   // a^((z - 1)^2)
    const F12 C = exp_z(B) * f12_frobenius(B);                              // ^(z + p)
    const F12 D = exp_z(exp_z(C)) * f12_frobenius(f12_frobenius(C)) * f12_conj(C);  // ^(z^2 + p^2 - 1)
    return D * f12_cyclotomic_sqr(a) * a;                                   // * a^3
}

// k * P for a canonical little-endian 256-bit scalar (4 x u64), 4-bit fixed windows
inline P1 p1_mul(const P1& p, const uint64_t k[4]) {
    P1 tab[16];
    tab[0] = p1_inf();
    tab[1] = p;
    for (int i = 2; i < 16; ++i) tab[i] = (i & 1) ? p1_add(tab[i - 1], p) : p1_double(tab[i / 2]);
    P1 acc = p1_inf();
    for (int i = 63; i >= 0; --i) {
        acc = p1_double(p1_double(p1_double(p1_double(acc))));
        const unsigned d = (unsigned)(k[i >> 4] >> (4 * (i & 15))) & 15u;
        if (d) acc = p1_add(acc, tab[d]);
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

// ---- G2 on the host: the one G2 element a verifier needs, setup_artifacts[1].g2 = [s]G2 ---------------------------
// (reference src/trusted_setup.rs:64-72 computes [s^k]G2 for every k with blst_p2_mult; only k = 1 is ever read,
// src/polynomial.rs:284).  Jacobian coordinates over Fp2, a = 0: dbl-2009-l / add-2007-bl, complete handling.
struct P2 {  // blst_p2 layout: x, y, z in Fp2 (a + b u, each Montgomery); z == 0 <=> infinity
    Fp2 x, y, z;
    bool is_inf() const { return z.a.is_zero() && z.b.is_zero(); }
};
static_assert(sizeof(P2) == 288, "blst_p2 is 36 x u64");
inline bool fp2_is_zero(const Fp2& x) { return x.a.is_zero() && x.b.is_zero(); }
inline P2 p2_inf() {
    P2 r;
    std::memset(&r, 0, sizeof r);
    return r;
}
inline P2 p2_double(const P2& p) {
    if (p.is_inf() || fp2_is_zero(p.y)) return p2_inf();
    const Fp2 A = fp2_sqr(p.x), B = fp2_sqr(p.y), C = fp2_sqr(B);
    const Fp2 t = fp2_sqr(p.x + B) - A - C, D = t + t, E = A + A + A, F = fp2_sqr(E);
    P2 r;
    r.x = F - D - D;
    Fp2 C8 = C + C;
    C8 = C8 + C8;
    C8 = C8 + C8;
    r.y = fp2_mul(E, D - r.x) - C8;
    const Fp2 yz = fp2_mul(p.y, p.z);
    r.z = yz + yz;
    return r;
}
inline P2 p2_add(const P2& a, const P2& b) {
    if (a.is_inf()) return b;
    if (b.is_inf()) return a;
    const Fp2 z1z1 = fp2_sqr(a.z), z2z2 = fp2_sqr(b.z);
    const Fp2 u1 = fp2_mul(a.x, z2z2), u2 = fp2_mul(b.x, z1z1);
    const Fp2 s1 = fp2_mul(fp2_mul(a.y, b.z), z2z2), s2 = fp2_mul(fp2_mul(b.y, a.z), z1z1);
    const Fp2 h = u2 - u1, rr = s2 - s1;
    if (fp2_is_zero(h)) return fp2_is_zero(rr) ? p2_double(a) : p2_inf();
    const Fp2 hh = fp2_sqr(h), hhh = fp2_mul(h, hh), v = fp2_mul(u1, hh);
    P2 r;
    r.x = fp2_sqr(rr) - hhh - v - v;
    r.y = fp2_mul(rr, v - r.x) - fp2_mul(s1, hhh);
    r.z = fp2_mul(fp2_mul(a.z, b.z), h);
    return r;
}
inline P2 p2_normalize(const P2& p) {  // z = 1
    if (p.is_inf()) return p2_inf();
    const Fp2 zi = fp2_inv(p.z), zi2 = fp2_sqr(zi);
    P2 r;
    r.x = fp2_mul(p.x, zi2);
    r.y = fp2_mul(fp2_mul(p.y, zi2), zi);
    r.z = fp2_one();
    return r;
}
// k * Q for a canonical little-endian 256-bit scalar, 4-bit fixed windows
inline P2 p2_mul(const P2& q, const uint64_t k[4]) {
    P2 tab[16];
    tab[0] = p2_inf();
    tab[1] = q;
    for (int i = 2; i < 16; ++i) tab[i] = (i & 1) ? p2_add(tab[i - 1], q) : p2_double(tab[i / 2]);
    P2 acc = p2_inf();
    for (int i = 63; i >= 0; --i) {
        acc = p2_double(p2_double(p2_double(p2_double(acc))));
        const unsigned d = (unsigned)(k[i >> 4] >> (4 * (i & 15))) & 15u;
        if (d) acc = p2_add(acc, tab[d]);
    }
    return acc;
}
inline P2 p2_generator() {
    const G2Affine g = g2_generator();
    P2 r;
    r.x = g.x;
    r.y = g.y;
    r.z = fp2_one();
    return r;
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
    const G2Affine qs[2] = {sg2, g2_generator()};
    const P1 ps[2] = {Pi, p1_neg(rhs)};
    bool ok = true;
    const F12 f = multi_miller_loop(qs, ps, 2, ok);
    if (!ok) return 0;
    return f12_is_one(f12_final_exp(f)) ? 1 : 0;
}

}  // namespace kzg_host
