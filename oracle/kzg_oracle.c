/* kzg_oracle.c -- CPU restatement (plain C, gcc) of the reference's commit / open path.
 *
 * TEST INFRASTRUCTURE ONLY -- see kzg_oracle.h for who may load it and for the
 * "parity unpinned" statement.  Each public function cites the reference file:line it follows.
 * The arithmetic the reference obtains from blst 0.3.15 (not vendored) is restated from the
 * published algorithms: word-serial Montgomery multiplication (CIOS), Jacobian addition
 * (add-2007-bl) / doubling (dbl-2009-l) with explicit exceptional cases, ZCash point compression.
 */
#include "kzg_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ generic n-limb helpers */
static inline __attribute__((always_inline)) uint64_t addn(uint64_t *o, const uint64_t *a, const uint64_t *b, int n) {
    u128 c = 0;
    for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; o[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline __attribute__((always_inline)) uint64_t subn(uint64_t *o, const uint64_t *a, const uint64_t *b, int n) {
    uint64_t br = 0;
    for (int i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        o[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
static inline int gen(const uint64_t *a, const uint64_t *b, int n) { /* a >= b */
    for (int i = n - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; }
    return 1;
}
static inline int zeron(const uint64_t *a, int n) {
    uint64_t o = 0;
    for (int i = 0; i < n; i++) o |= a[i];
    return o == 0;
}
/* Montgomery product, CIOS, n limbs, result fully reduced */
static inline __attribute__((always_inline)) void montmul(uint64_t *o, const uint64_t *a, const uint64_t *b, const uint64_t *m, uint64_t n0, int n) {
    uint64_t t[8] = {0};
    for (int i = 0; i < n; i++) {
        u128 c = 0;
        for (int j = 0; j < n; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[n]; t[n] = (uint64_t)c; t[n + 1] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * n0;
        c = (u128)q * m[0] + t[0]; c >>= 64;
        for (int j = 1; j < n; j++) { c += (u128)q * m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[n]; t[n - 1] = (uint64_t)c; t[n] = t[n + 1] + (uint64_t)(c >> 64);
    }
    if (t[n] || gen(t, m, n)) subn(o, t, m, n); else memcpy(o, t, 8 * n);
}

/* ------------------------------------------------------------------ Fp */
static const ofp FP_P = {{0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL}};
static const ofp FP_ONE = {{0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL}};
static const ofp FP_R2 = {{0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL}};
static const uint64_t FP_N0 = 0x89f3fffcfffcfffdULL;
static const ofp FP_HALF = {{0xdcff7fffffffd555ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL, 0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL}}; /* (p-1)/2 */
static const ofp FP_PM2 = {{0xb9feffffffffaaa9ULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL}};
static const ofp FP_SQRT_E = {{0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL, 0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL}}; /* (p+1)/4 */
static const ofp FP_B = {{0xaa270000000cfff3ULL, 0x53cc0032fc34000aULL, 0x478fe97a6b0a807fULL, 0xb1d37ebee6ba24d7ULL, 0x8ec9733bbf78ab2fULL, 0x09d645513d83de7eULL}}; /* 4, Montgomery */
static const ofp G1_X = {{0x5cb38790fd530c16ULL, 0x7817fc679976fff5ULL, 0x154f95c7143ba1c1ULL, 0xf0ae6acdf3d0e747ULL, 0xedce6ecc21dbf440ULL, 0x120177419e0bfb75ULL}};
static const ofp G1_Y = {{0xbaac93d50ce72271ULL, 0x8c22631a7918fd8eULL, 0xdd595f13570725ceULL, 0x51ac582950405194ULL, 0x0e1c8c3fad0059c0ULL, 0x0bbc3efc5008a26aULL}};

static inline void fp_mul(ofp *o, const ofp *a, const ofp *b) { montmul(o->l, a->l, b->l, FP_P.l, FP_N0, 6); }
static inline void fp_sqr(ofp *o, const ofp *a) { fp_mul(o, a, a); }
static inline void fp_add(ofp *o, const ofp *a, const ofp *b) {
    uint64_t t[6];
    addn(t, a->l, b->l, 6);                        /* p < 2^381: no carry out of 384 bits */
    if (gen(t, FP_P.l, 6)) subn(o->l, t, FP_P.l, 6); else memcpy(o->l, t, 48);
}
static inline void fp_sub(ofp *o, const ofp *a, const ofp *b) {
    uint64_t t[6];
    if (subn(t, a->l, b->l, 6)) addn(t, t, FP_P.l, 6);
    memcpy(o->l, t, 48);
}
static inline void fp_neg(ofp *o, const ofp *a) {
    if (zeron(a->l, 6)) { memset(o, 0, sizeof *o); return; }
    subn(o->l, FP_P.l, a->l, 6);
}
static inline int fp_is_zero(const ofp *a) { return zeron(a->l, 6); }
static inline int fp_eq(const ofp *a, const ofp *b) { return memcmp(a, b, sizeof *a) == 0; }
static void fp_pow(ofp *o, const ofp *a, const ofp *e) {
    ofp acc = FP_ONE, base = *a;
    for (int i = 0; i < 384; i++) {
        if ((e->l[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, &base);
        fp_sqr(&base, &base);
    }
    *o = acc;
}
static void fp_inv(ofp *o, const ofp *a) { fp_pow(o, a, &FP_PM2); }
static void fp_from_mont(ofp *o, const ofp *a) { ofp one = {{1, 0, 0, 0, 0, 0}}; fp_mul(o, a, &one); }
static void fp_to_mont(ofp *o, const ofp *a) { fp_mul(o, a, &FP_R2); }

/* ------------------------------------------------------------------ Fr  (reference src/scalar.rs) */
static const ofr FR_MOD = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}};
static const ofr FR_ONE = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};
static const ofr FR_R2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}};
static const uint64_t FR_N0 = 0xfffffffeffffffffULL;

void oracle_fr_mul(ofr *o, const ofr *a, const ofr *b) { montmul(o->l, a->l, b->l, FR_MOD.l, FR_N0, 4); }
void oracle_fr_add(ofr *o, const ofr *a, const ofr *b) {
    uint64_t t[4];
    uint64_t c = addn(t, a->l, b->l, 4);
    if (c || gen(t, FR_MOD.l, 4)) subn(o->l, t, FR_MOD.l, 4); else memcpy(o->l, t, 32);
}
void oracle_fr_sub(ofr *o, const ofr *a, const ofr *b) {
    uint64_t t[4];
    if (subn(t, a->l, b->l, 4)) addn(t, t, FR_MOD.l, 4);
    memcpy(o->l, t, 32);
}
void oracle_fr_neg(ofr *o, const ofr *a) {
    if (zeron(a->l, 4)) { memset(o, 0, sizeof *o); return; }
    subn(o->l, FR_MOD.l, a->l, 4);
}
int oracle_fr_is_zero(const ofr *a) { return zeron(a->l, 4); }

/* raw 256-bit integer (LE limbs, any value < 2^256) -> Montgomery Fr, reducing mod r, as
 * blst_fr_from_hexascii does for the reference's constructors (scalar.rs:34, 58, 70) */
static void fr_from_raw(ofr *o, const uint64_t raw[4]) {
    ofr t;
    memcpy(t.l, raw, 32);
    /* montmul accepts any a < 2^256 with b < r: result is a*b/R mod r, fully reduced */
    oracle_fr_mul(o, &t, &FR_R2);
}
void oracle_fr_from_le_bytes(ofr *o, const uint8_t b[32]) {
    uint64_t raw[4];
    for (int i = 0; i < 4; i++) { raw[i] = 0; for (int k = 7; k >= 0; k--) raw[i] = (raw[i] << 8) | b[8 * i + k]; }
    fr_from_raw(o, raw);
}
void oracle_fr_from_be_bytes(ofr *o, const uint8_t b[32]) {
    uint8_t le[32];
    for (int i = 0; i < 32; i++) le[i] = b[31 - i];
    oracle_fr_from_le_bytes(o, le);
}
void oracle_fr_from_i128(ofr *o, int64_t hi, uint64_t lo) {
    /* value = hi*2^64 + lo as a two's complement i128.  scalar.rs:27-48 */
    int neg = hi < 0;
    uint64_t ahi = (uint64_t)hi, alo = lo;
    if (neg) { alo = ~alo + 1; ahi = ~ahi + (alo == 0); }      /* unsigned_abs */
    uint64_t raw[4] = {alo, ahi, 0, 0};
    ofr mag;
    fr_from_raw(&mag, raw);
    int positive = !neg && (ahi | alo);
    if (positive) { *o = mag; return; }
    ofr zero = {{0, 0, 0, 0}};
    oracle_fr_sub(o, &zero, &mag);                              /* r - |a|  (0 stays 0) */
}
void oracle_fr_to_le_bytes(uint8_t out[32], const ofr *a) {
    ofr one = {{1, 0, 0, 0}}, c;
    oracle_fr_mul(&c, a, &one);                                 /* leave Montgomery form */
    for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(c.l[i] >> (8 * k));
}
void oracle_fr_pow(ofr *o, const ofr *a, uint64_t e) {
    ofr acc = FR_ONE, base = *a;
    while (e) { if (e & 1) oracle_fr_mul(&acc, &acc, &base); oracle_fr_mul(&base, &base, &base); e >>= 1; }
    *o = acc;
}

/* ------------------------------------------------------------------ G1 (reference src/curves.rs) */
int oracle_p1_is_inf(const op1 *p) { return fp_is_zero(&p->z); }
void oracle_p1_generator(op1 *o) { o->x = G1_X; o->y = G1_Y; o->z = FP_ONE; }
void oracle_p1_cneg(op1 *a, int flag) { if (flag) fp_neg(&a->y, &a->y); }

void oracle_p1_double(op1 *o, const op1 *p) { /* dbl-2009-l, a = 0 */
    if (oracle_p1_is_inf(p) || fp_is_zero(&p->y)) { memset(o, 0, sizeof *o); return; }
    ofp A, B, C, D, E, F, t, X3, Y3, Z3;
    fp_sqr(&A, &p->x); fp_sqr(&B, &p->y); fp_sqr(&C, &B);
    fp_add(&t, &p->x, &B); fp_sqr(&t, &t); fp_sub(&t, &t, &A); fp_sub(&t, &t, &C); fp_add(&D, &t, &t);
    fp_add(&E, &A, &A); fp_add(&E, &E, &A);
    fp_sqr(&F, &E);
    fp_sub(&X3, &F, &D); fp_sub(&X3, &X3, &D);
    fp_sub(&t, &D, &X3); fp_mul(&Y3, &E, &t);
    fp_add(&C, &C, &C); fp_add(&C, &C, &C); fp_add(&C, &C, &C); fp_sub(&Y3, &Y3, &C);
    fp_mul(&Z3, &p->y, &p->z); fp_add(&Z3, &Z3, &Z3);
    o->x = X3; o->y = Y3; o->z = Z3;
}

void oracle_p1_add_or_double(op1 *o, const op1 *a, const op1 *b) { /* curves.rs:79-85 semantics */
    if (oracle_p1_is_inf(a)) { *o = *b; return; }
    if (oracle_p1_is_inf(b)) { *o = *a; return; }
    ofp Z1Z1, Z2Z2, U1, U2, S1, S2, H, Rr, t;
    fp_sqr(&Z1Z1, &a->z); fp_sqr(&Z2Z2, &b->z);
    fp_mul(&U1, &a->x, &Z2Z2); fp_mul(&U2, &b->x, &Z1Z1);
    fp_mul(&t, &b->z, &Z2Z2); fp_mul(&S1, &a->y, &t);
    fp_mul(&t, &a->z, &Z1Z1); fp_mul(&S2, &b->y, &t);
    fp_sub(&H, &U2, &U1); fp_sub(&Rr, &S2, &S1);
    if (fp_is_zero(&H)) {
        if (fp_is_zero(&Rr)) { oracle_p1_double(o, a); return; }
        memset(o, 0, sizeof *o); return;
    }
    ofp HH, HHH, V, X3, Y3, Z3;
    fp_sqr(&HH, &H); fp_mul(&HHH, &H, &HH); fp_mul(&V, &U1, &HH);
    fp_sqr(&X3, &Rr); fp_sub(&X3, &X3, &HHH); fp_sub(&X3, &X3, &V); fp_sub(&X3, &X3, &V);
    fp_sub(&t, &V, &X3); fp_mul(&Y3, &Rr, &t); fp_mul(&t, &S1, &HHH); fp_sub(&Y3, &Y3, &t);
    fp_mul(&Z3, &a->z, &b->z); fp_mul(&Z3, &Z3, &H);
    o->x = X3; o->y = Y3; o->z = Z3;
}

void oracle_p1_mult(op1 *o, const op1 *p, const uint8_t *s, size_t nbits) {
    /* curves.rs:90-96: blst_p1_mult(out, p, scalar_le, 256).  Fixed 4-bit windows, MSB first. */
    op1 tab[16];
    memset(&tab[0], 0, sizeof tab[0]);
    tab[1] = *p;
    for (int i = 2; i < 16; i++) oracle_p1_add_or_double(&tab[i], &tab[i - 1], p);
    op1 acc;
    memset(&acc, 0, sizeof acc);
    size_t nn = (nbits + 3) / 4;
    for (size_t w = nn; w-- > 0;) {
        for (int k = 0; k < 4; k++) oracle_p1_double(&acc, &acc);
        size_t bit = 4 * w;
        unsigned d = 0;
        for (int k = 3; k >= 0; k--) {
            size_t bi = bit + k;
            unsigned v = bi < nbits ? (s[bi >> 3] >> (bi & 7)) & 1 : 0;
            d = (d << 1) | v;
        }
        if (d) oracle_p1_add_or_double(&acc, &acc, &tab[d]);
    }
    *o = acc;
}

void oracle_p1_to_affine(op1 *o, const op1 *p) {
    if (oracle_p1_is_inf(p)) { memset(o, 0, sizeof *o); return; }
    ofp zi, zi2, zi3;
    fp_inv(&zi, &p->z); fp_sqr(&zi2, &zi); fp_mul(&zi3, &zi2, &zi);
    fp_mul(&o->x, &p->x, &zi2); fp_mul(&o->y, &p->y, &zi3); o->z = FP_ONE;
}
void oracle_p1_rescale(op1 *o, const op1 *p, const ofp *lam) {
    ofp l2, l3;
    fp_sqr(&l2, lam); fp_mul(&l3, &l2, lam);
    fp_mul(&o->x, &p->x, &l2); fp_mul(&o->y, &p->y, &l3); fp_mul(&o->z, &p->z, lam);
}
int oracle_p1_on_curve(const op1 *p) {
    if (oracle_p1_is_inf(p)) return 1;
    op1 a; oracle_p1_to_affine(&a, p);
    ofp l, r;
    fp_sqr(&l, &a.y); fp_sqr(&r, &a.x); fp_mul(&r, &r, &a.x); fp_add(&r, &r, &FP_B);
    return fp_eq(&l, &r);
}
int oracle_p1_equal(const op1 *a, const op1 *b) {
    int ia = oracle_p1_is_inf(a), ib = oracle_p1_is_inf(b);
    if (ia || ib) return ia && ib;
    ofp za2, zb2, za3, zb3, l, r;
    fp_sqr(&za2, &a->z); fp_sqr(&zb2, &b->z);
    fp_mul(&l, &a->x, &zb2); fp_mul(&r, &b->x, &za2);
    if (!fp_eq(&l, &r)) return 0;
    fp_mul(&za3, &za2, &a->z); fp_mul(&zb3, &zb2, &b->z);
    fp_mul(&l, &a->y, &zb3); fp_mul(&r, &b->y, &za3);
    return fp_eq(&l, &r);
}

void oracle_p1_compress(uint8_t out[48], const op1 *p) { /* curves.rs:99-110 -> blst_p1_compress */
    if (oracle_p1_is_inf(p)) { memset(out, 0, 48); out[0] = 0xC0; return; }
    op1 a; oracle_p1_to_affine(&a, p);
    ofp x, y;
    fp_from_mont(&x, &a.x); fp_from_mont(&y, &a.y);
    for (int i = 0; i < 6; i++) for (int k = 0; k < 8; k++) out[47 - (8 * i + k)] = (uint8_t)(x.l[i] >> (8 * k));
    out[0] |= 0x80;
    /* sign bit: y > (p-1)/2 */
    int larger = 0;
    for (int i = 5; i >= 0; i--) { if (y.l[i] != FP_HALF.l[i]) { larger = y.l[i] > FP_HALF.l[i]; break; } }
    if (larger) out[0] |= 0x20;
}
int oracle_p1_uncompress(op1 *o, const uint8_t in[48]) { /* curves.rs:131-142 */
    if (!(in[0] & 0x80)) return -1;
    if (in[0] & 0x40) { memset(o, 0, sizeof *o); return 0; }
    uint8_t b[48]; memcpy(b, in, 48); b[0] &= 0x1F;
    ofp x; memset(&x, 0, sizeof x);
    for (int i = 0; i < 48; i++) x.l[(47 - i) >> 3] |= (uint64_t)b[i] << (8 * ((47 - i) & 7));
    if (gen(x.l, FP_P.l, 6)) return -2;
    ofp xm, y2, y, chk, yc;
    fp_to_mont(&xm, &x);
    fp_sqr(&y2, &xm); fp_mul(&y2, &y2, &xm); fp_add(&y2, &y2, &FP_B);
    fp_pow(&y, &y2, &FP_SQRT_E);
    fp_sqr(&chk, &y);
    if (!fp_eq(&chk, &y2)) return -3;
    fp_from_mont(&yc, &y);
    int larger = 0;
    for (int i = 5; i >= 0; i--) { if (yc.l[i] != FP_HALF.l[i]) { larger = yc.l[i] > FP_HALF.l[i]; break; } }
    if (larger != !!(in[0] & 0x20)) fp_neg(&y, &y);
    o->x = xm; o->y = y; o->z = FP_ONE;
    return 0;
}

/* ------------------------------------------------------------------ trusted setup (trusted_setup.rs) */
void oracle_srs_g1(op1 *out, size_t n, const uint8_t secret_be[32]) {
    ofr s, cur = FR_ONE;
    oracle_fr_from_be_bytes(&s, secret_be);                       /* :24 */
    op1 g; oracle_p1_generator(&g);
    for (size_t k = 0; k < n; k++) {
        if (k == 0) { out[0] = g; continue; }                     /* :41-48 */
        oracle_fr_mul(&cur, &cur, &s);                            /* :50 */
        uint8_t le[32]; oracle_fr_to_le_bytes(le, &cur);          /* :52 */
        oracle_p1_mult(&out[k], &g, le, 256);                     /* :54-62 */
    }
}
void oracle_srs_g1_at(op1 *out, uint64_t k, const uint8_t secret_be[32]) {
    ofr s, sk;
    oracle_fr_from_be_bytes(&s, secret_be);
    oracle_fr_pow(&sk, &s, k);
    uint8_t le[32]; oracle_fr_to_le_bytes(le, &sk);
    op1 g; oracle_p1_generator(&g);
    oracle_p1_mult(out, &g, le, 256);
}

/* ------------------------------------------------------------------ polynomial path (polynomial.rs) */
size_t oracle_poly_truncate(const ofr *c, size_t n) { /* :55-75 */
    size_t last = 0;
    for (size_t i = 0; i < n; i++) if (!oracle_fr_is_zero(&c[i])) last = i;
    return n == 0 ? 0 : last + 1;
}
void oracle_poly_evaluate(ofr *out, const ofr *c, size_t n, const ofr *x) { /* :112-123 */
    ofr acc; memset(&acc, 0, sizeof acc);
    for (size_t i = n; i-- > 0;) { oracle_fr_mul(&acc, &acc, x); oracle_fr_add(&acc, &acc, &c[i]); }
    *out = acc;
}
static inline const op1 *srs_at(const void *first, size_t stride, size_t i) {
    return (const op1 *)((const uint8_t *)first + i * stride);
}
int oracle_commit_naive(op1 *out, const ofr *c, size_t n, const void *srs, size_t stride, size_t srs_len) {
    size_t degree = n == 0 ? 0 : n - 1;                           /* :93-98 */
    if (degree + 1 > srs_len) return ORACLE_ERR_DEGREE_TOO_HIGH;  /* :201-205 */
    op1 acc; memset(&acc, 0, sizeof acc);                         /* :207 G1Point::from_i128(0) */
    for (size_t i = 0; i < n; i++) {                              /* :208-212 */
        uint8_t le[32]; oracle_fr_to_le_bytes(le, &c[i]);
        op1 t; oracle_p1_mult(&t, srs_at(srs, stride, i), le, 256);
        oracle_p1_add_or_double(&acc, &acc, &t);
    }
    *out = acc;
    return ORACLE_OK;
}
int oracle_quotient(ofr *q, size_t *qn, const ofr *c, size_t n, const ofr *z, const ofr *y) {
    /* (P - y): polynomial.rs:128-145 with other = from_constant(y) (:78-89), then :150-195 */
    *qn = 0;
    ofr c0; memset(&c0, 0, sizeof c0);
    size_t m;                                                     /* length of P - y after truncation */
    if (n > 1) { oracle_fr_sub(&c0, &c[0], y); m = n; }           /* a_len > b_len (b_len <= 1) */
    else if (n == 1 && oracle_fr_is_zero(y)) { c0 = c[0]; m = 1; }/* b = [] : a_len(1) > 0 */
    else if (n == 1) { oracle_fr_sub(&c0, &c[0], y); m = 1; }     /* else-branch: -y + c0 */
    else { if (oracle_fr_is_zero(y)) m = 0; else { oracle_fr_neg(&c0, y); m = 1; } } /* [] - [y] */
    /* TryFrom truncation (:55-75): only c0 changed, so trailing zeros are those of c itself */
    if (m > 1) { size_t last = 0; for (size_t i = 1; i < m; i++) if (!oracle_fr_is_zero(&c[i])) last = i; m = last + 1; }
    if (m == 0) return ORACLE_OK;                                 /* :151-157 */
    if (m == 1) return oracle_fr_is_zero(&c0) ? ORACLE_OK : ORACLE_ERR_CONSTANT_POLY; /* :159-167 */
    size_t d = m - 1;
    q[d - 1] = c[d];                                              /* :168 */
    for (size_t i = d - 1; i >= 1; i--) {                         /* :171-179 */
        ofr t; oracle_fr_mul(&t, z, &q[i]); oracle_fr_add(&q[i - 1], &c[i], &t);
    }
    ofr rb; oracle_fr_mul(&rb, z, &q[0]); oracle_fr_neg(&rb, &rb);/* :184-186 */
    if (memcmp(&rb, &c0, sizeof rb) != 0) return ORACLE_ERR_REMAINDER; /* :188-192 */
    *qn = oracle_poly_truncate(q, d);                             /* :194 */
    return ORACLE_OK;
}
int oracle_generate_proof(op1 *out, const ofr *c, size_t n, const ofr *z, const ofr *y,
                          const void *srs, size_t stride, size_t srs_len) { /* :260-269 */
    ofr *q = (ofr *)malloc(sizeof(ofr) * (n ? n : 1));
    size_t qn = 0;
    int rc = oracle_quotient(q, &qn, c, n, z, y);
    if (rc == ORACLE_OK) rc = oracle_commit_naive(out, q, qn, srs, stride, srs_len);
    free(q);
    return rc;
}
void oracle_commit_shortcut(op1 *out, const ofr *c, size_t n, const uint8_t secret_be[32]) {
    ofr s, v;
    oracle_fr_from_be_bytes(&s, secret_be);
    oracle_poly_evaluate(&v, c, n, &s);
    uint8_t le[32]; oracle_fr_to_le_bytes(le, &v);
    op1 g; oracle_p1_generator(&g);
    oracle_p1_mult(out, &g, le, 256);
}

void oracle_bench_coefficients(ofr *out, size_t n) { /* benches/polynomial_commitment.rs:10-15 */
    ofr five, ten, p5 = FR_ONE;
    oracle_fr_from_i128(&five, 0, 5); oracle_fr_from_i128(&ten, 0, 10);
    for (size_t i = 0; i < n; i++) { oracle_fr_add(&out[i], &p5, &ten); oracle_fr_mul(&p5, &p5, &five); }
}
void oracle_bench_input_point(ofr *out, uint64_t degree) { /* benches/evaluation_proof.rs:25-27 */
    ofr five, twenty, p;
    oracle_fr_from_i128(&five, 0, 5); oracle_fr_from_i128(&twenty, 0, 20);
    oracle_fr_pow(&p, &five, degree);
    oracle_fr_add(out, &p, &twenty);
}

/* ------------------------------------------------------------------ CPU bucket MSM (strong baseline)
 * Unsigned c-bit windows, Jacobian buckets; the work is cut into (window, point range) jobs over pthreads so that
 * every host core gets one (windows alone are ~19 jobs).  Computes the same group element as oracle_commit_naive; it
 * is a baseline and a cross-check, not a restatement. */
typedef struct {
    const ofr *c; size_t i0, i1; const void *srs; size_t stride;
    const uint8_t *canon; int cbits; int w; op1 *partial;
} pip_job;
typedef struct {
    pip_job *jobs; int njobs; int next; pthread_mutex_t mu;
} pip_queue;

static void pip_run(pip_job *J, op1 *bk) {
    size_t nb = ((size_t)1 << J->cbits) - 1;
    memset(bk, 0, sizeof(op1) * nb);
    for (size_t i = J->i0; i < J->i1; i++) {
        const uint8_t *s = J->canon + 32 * i;
        size_t bit = (size_t)J->w * J->cbits;
        unsigned d = 0;
        for (int k = J->cbits - 1; k >= 0; k--) {
            size_t bi = bit + k;
            unsigned v = bi < 256 ? (s[bi >> 3] >> (bi & 7)) & 1 : 0;
            d = (d << 1) | v;
        }
        if (d) oracle_p1_add_or_double(&bk[d - 1], &bk[d - 1], srs_at(J->srs, J->stride, i));
    }
    op1 run, acc;
    memset(&run, 0, sizeof run); memset(&acc, 0, sizeof acc);
    for (size_t b = nb; b-- > 0;) {
        oracle_p1_add_or_double(&run, &run, &bk[b]);
        oracle_p1_add_or_double(&acc, &acc, &run);
    }
    *J->partial = acc;
}
static void *pip_worker(void *arg) {
    pip_queue *Q = (pip_queue *)arg;
    op1 *bk = NULL;
    for (;;) {
        pthread_mutex_lock(&Q->mu);
        int k = Q->next < Q->njobs ? Q->next++ : -1;
        pthread_mutex_unlock(&Q->mu);
        if (k < 0) break;
        if (!bk) bk = (op1 *)malloc(sizeof(op1) * (((size_t)1 << Q->jobs[k].cbits) - 1));
        pip_run(&Q->jobs[k], bk);
    }
    free(bk);
    return NULL;
}

/* threads_used (may be NULL) receives the number of threads that had a job */
int oracle_commit_pippenger_ex(op1 *out, const ofr *c, size_t n, const void *srs, size_t stride,
                               size_t srs_len, int threads, int *threads_used) {
    size_t degree = n == 0 ? 0 : n - 1;
    if (degree + 1 > srs_len) return ORACLE_ERR_DEGREE_TOO_HIGH;
    memset(out, 0, sizeof *out);
    if (threads_used) *threads_used = 0;
    if (n == 0) return ORACLE_OK;
    if (threads < 1) threads = 1;
    /* window width and point ranges: a job costs (points of its range) + 2 * 2^c additions; the jobs run `threads` at a time */
    int cbits = 4, ranges = 1;
    double best = 1e300;
    for (int cb = 2; cb <= 16; cb++) {
        int nw = (255 + cb - 1) / cb;
        int r = threads / nw;
        if (r < 1) r = 1;
        if ((size_t)r > n) r = (int)n;
        double per_job = (double)n / r + 2.0 * (double)((size_t)1 << cb);
        double rounds = (double)((nw * r + threads - 1) / threads);
        if (per_job * rounds < best) { best = per_job * rounds; cbits = cb; ranges = r; }
    }
    int nwin = (255 + cbits - 1) / cbits;
    int njobs = nwin * ranges;
    uint8_t *canon = (uint8_t *)malloc(32 * n);
    for (size_t i = 0; i < n; i++) oracle_fr_to_le_bytes(canon + 32 * i, &c[i]);
    op1 *partial = (op1 *)malloc(sizeof(op1) * njobs);
    pip_job *jobs = (pip_job *)malloc(sizeof(pip_job) * njobs);
    for (int w = 0; w < nwin; w++)
        for (int r = 0; r < ranges; r++)
            jobs[w * ranges + r] = (pip_job){c, n * r / ranges, n * (r + 1) / ranges, srs, stride, canon, cbits, w, &partial[w * ranges + r]};
    if (threads > njobs) threads = njobs;
    pip_queue Q = {jobs, njobs, 0, PTHREAD_MUTEX_INITIALIZER};
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, pip_worker, &Q);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    if (threads_used) *threads_used = threads;
    op1 acc; memset(&acc, 0, sizeof acc);
    for (int w = nwin; w-- > 0;) {
        for (int k = 0; k < cbits; k++) oracle_p1_double(&acc, &acc);
        for (int r = 0; r < ranges; r++) oracle_p1_add_or_double(&acc, &acc, &partial[w * ranges + r]);
    }
    *out = acc;
    free(jobs); free(th); free(partial); free(canon);
    return ORACLE_OK;
}
int oracle_commit_pippenger(op1 *out, const ofr *c, size_t n, const void *srs, size_t stride,
                            size_t srs_len, int threads) {
    return oracle_commit_pippenger_ex(out, c, n, srs, stride, srs_len, threads, NULL);
}
