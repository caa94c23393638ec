/* kzg_oracle.h -- CPU restatement of the reference's commit / open path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product
 * (kzg_poly_commit_exploration_amd/) never links, imports or calls it.
 *
 * Parity status: "parity unpinned" for G1 outputs -- the reference
 * (VGLoic/kzg-poly-commit-exploration @ 2025-09-19) performs all arithmetic through the
 * un-vendored crate blst 0.3.15 (Cargo.toml:10, Cargo.lock:89-92), cannot be built here (no
 * cargo/rustc, no network) and holds no known-answer commitment or proof bytes.  The oracle is
 * pinned by: public BLS12-381 constants and the public ZCash encodings of G and 2G; the
 * reference's own Fr byte-semantics tests (src/scalar.rs:350-389); and three-way triangulation
 * (this file / oracle/bigint_twin.py / the [P(s)]G shortcut) -- see tests/test_oracle.py.
 *
 * Memory layouts are blst's: blst_fr = 4 x u64 LE limbs, Montgomery (R = 2^256);
 * blst_fp = 6 x u64 LE limbs, Montgomery (R = 2^384); blst_p1 = {x,y,z} Jacobian, z == 0 <=> inf.
 */
#ifndef KZG_ORACLE_H
#define KZG_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } ofr;          /* blst_fr  */
typedef struct { uint64_t l[6]; } ofp;          /* blst_fp  */
typedef struct { ofp x, y, z; } op1;            /* blst_p1  */

/* error codes mirror the reference's anyhow messages (src/polynomial.rs:165, 189-191, 202-204) */
#define ORACLE_OK 0
#define ORACLE_ERR_DEGREE_TOO_HIGH (-1)
#define ORACLE_ERR_CONSTANT_POLY (-2)
#define ORACLE_ERR_REMAINDER (-3)

/* ---- Fr: reference src/scalar.rs ---- */
void oracle_fr_from_le_bytes(ofr *out, const uint8_t b[32]);   /* scalar.rs:54-61 (reduces mod r) */
void oracle_fr_from_be_bytes(ofr *out, const uint8_t b[32]);   /* scalar.rs:66-73 */
void oracle_fr_from_i128(ofr *out, int64_t hi, uint64_t lo);   /* scalar.rs:27-48: a>0 -> a, a<=0 -> r-|a| */
void oracle_fr_to_le_bytes(uint8_t out[32], const ofr *a);     /* scalar.rs:83-93 */
void oracle_fr_mul(ofr *out, const ofr *a, const ofr *b);      /* scalar.rs:111-117 */
void oracle_fr_add(ofr *out, const ofr *a, const ofr *b);      /* scalar.rs:192-198 */
void oracle_fr_sub(ofr *out, const ofr *a, const ofr *b);      /* scalar.rs:203-209 */
void oracle_fr_neg(ofr *out, const ofr *a);                    /* scalar.rs:212-218 */
int oracle_fr_is_zero(const ofr *a);                           /* scalar.rs:221-223 */
void oracle_fr_pow(ofr *out, const ofr *a, uint64_t e);        /* scalar.rs:122-187 (same value) */

/* ---- G1: reference src/curves.rs ---- */
void oracle_p1_generator(op1 *out);                                   /* blst_p1_generator */
void oracle_p1_add_or_double(op1 *out, const op1 *a, const op1 *b);   /* curves.rs:79-85 */
void oracle_p1_double(op1 *out, const op1 *a);
void oracle_p1_cneg(op1 *a, int flag);                                /* curves.rs:41,70 */
void oracle_p1_mult(op1 *out, const op1 *p, const uint8_t *scalar_le, size_t nbits); /* curves.rs:90-96 */
void oracle_p1_compress(uint8_t out[48], const op1 *p);               /* curves.rs:99-110 */
int oracle_p1_uncompress(op1 *out, const uint8_t in[48]);             /* curves.rs:131-142; 0 = ok */
int oracle_p1_is_inf(const op1 *p);
int oracle_p1_on_curve(const op1 *p);
int oracle_p1_equal(const op1 *a, const op1 *b);                      /* same group element */
void oracle_p1_to_affine(op1 *out, const op1 *p);                     /* z := 1 (Montgomery R) or inf */
void oracle_p1_rescale(op1 *out, const op1 *p, const ofp *lambda_mont); /* (X l^2, Y l^3, Z l): same point, other Z */

/* ---- trusted setup, G1 side: reference src/trusted_setup.rs:20-28, 40-62 ---- */
void oracle_srs_g1(op1 *out, size_t n, const uint8_t secret_be[32]);
/* single entry [s^k]G without walking the whole sequence (spot checks at large k) */
void oracle_srs_g1_at(op1 *out, uint64_t k, const uint8_t secret_be[32]);

/* ---- polynomial path: reference src/polynomial.rs ---- */
size_t oracle_poly_truncate(const ofr *c, size_t n);                     /* :55-75, returns new len */
void oracle_poly_evaluate(ofr *out, const ofr *c, size_t n, const ofr *x); /* :112-123 (value) */
/* :200-215, naive N scalar-muls + N adds, single thread.  srs strided like &srs[0].g1 */
int oracle_commit_naive(op1 *out, const ofr *c, size_t n, const void *srs_first, size_t stride, size_t srs_len);
/* :150-195 on (P - y), i.e. sub (:128-145) then divide.  q must hold n-1 entries; *qn = truncated length */
int oracle_quotient(ofr *q, size_t *qn, const ofr *c, size_t n, const ofr *z, const ofr *y);
/* :260-269 */
int oracle_generate_proof(op1 *out, const ofr *c, size_t n, const ofr *z, const ofr *y,
                          const void *srs_first, size_t stride, size_t srs_len);
/* bucket-method MSM, `threads` pthreads: the strong CPU baseline; same group element as commit_naive */
int oracle_commit_pippenger(op1 *out, const ofr *c, size_t n, const void *srs_first, size_t stride,
                            size_t srs_len, int threads);
/* the same; *threads_used receives how many threads had work (jobs are (window, point range) pairs) */
int oracle_commit_pippenger_ex(op1 *out, const ofr *c, size_t n, const void *srs, size_t stride,
                               size_t srs_len, int threads, int *threads_used);
/* shortcut valid when the secret is known: [P(s)]G   */
void oracle_commit_shortcut(op1 *out, const ofr *c, size_t n, const uint8_t secret_be[32]);

/* bench input generators: reference benches/polynomial_commitment.rs:10-15, evaluation_proof.rs:25-27 */
void oracle_bench_coefficients(ofr *out, size_t n);       /* c_i = 5^i + 10 */
void oracle_bench_input_point(ofr *out, uint64_t degree); /* z = 5^d + 20 */

#ifdef __cplusplus
}
#endif
#endif
