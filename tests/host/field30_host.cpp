// Host build of csrc/field30.hip.h for tests/test_field30.py (plain g++; the header is __host__ __device__ code).
// Test infrastructure only.  Exposes the digit-level routines over a C ABI and a variant of the multiplier that
// reports the largest column magnitude (to check the "fits a signed 64-bit accumulator" contract).
#include <stdint.h>
#include <string.h>

#include "../../kzg_poly_commit_exploration_amd/csrc/field30_inv.hip.h"
#include "../../kzg_poly_commit_exploration_amd/csrc/g1_30.hip.h"
#include "../../kzg_poly_commit_exploration_amd/csrc/host_field.hpp"

using namespace kzg;

extern "C" {

void f30_mul(const int32_t* a, const int32_t* b, int32_t* r) {
    Fq x, y;
    memcpy(x.d, a, sizeof x.d);
    memcpy(y.d, b, sizeof y.d);
    Fq z = fq_mul(x, y);
    memcpy(r, z.d, sizeof z.d);
}
void f30_mul_sub(const int32_t* a, const int32_t* b, const int32_t* c, const int32_t* d, int32_t* r) {
    Fq x, y, z, w;
    memcpy(x.d, a, sizeof x.d);
    memcpy(y.d, b, sizeof y.d);
    memcpy(z.d, c, sizeof z.d);
    memcpy(w.d, d, sizeof w.d);
    Fq o = fq_mul_sub(x, y, z, w);
    memcpy(r, o.d, sizeof o.d);
}
// largest |column| / 2^48 of fq_mul_sub in exact arithmetic (sum of magnitudes: the worst sign pattern)
int64_t f30_mul_sub_max_column(const int32_t* a, const int32_t* b, const int32_t* c, const int32_t* d) {
    __int128 worst = 0;
    for (int k = 0; k < 25; k++) {
        __int128 mag = (__int128)1 << 34;  // carry in
        for (int i = 0; i < 13; i++) {
            int j = k - i;
            if (j < 0 || j > 12) continue;
            __int128 t = (__int128)a[i] * b[j], u = (__int128)c[i] * d[j];
            mag += (t < 0 ? -t : t) + (u < 0 ? -u : u);
            __int128 pm = (__int128)(1 << 29) * fq_pd(j);  // |m_i| <= 2^29 times |p_j|
            mag += pm < 0 ? -pm : pm;
        }
        if (mag > worst) worst = mag;
    }
    return (int64_t)(worst >> 48);
}
void f30_sqr(const int32_t* a, int32_t* r) {
    Fq x;
    memcpy(x.d, a, sizeof x.d);
    Fq z = fq_sqr(x);
    memcpy(r, z.d, sizeof z.d);
}
void f30_norm(const int32_t* a, int32_t* r) {
    Fq x;
    memcpy(x.d, a, sizeof x.d);
    Fq z = fq_norm(x);
    memcpy(r, z.d, sizeof z.d);
}
int f30_is_zero(const int32_t* a) {
    Fq x;
    memcpy(x.d, a, sizeof x.d);
    return fq_is_zero(x) ? 1 : 0;
}
void f30_from_u32x12(const uint32_t* s, int32_t* r) {
    Fq z = fq_from_u32x12(s);
    memcpy(r, z.d, sizeof z.d);
}
void f30_to_u32x12(const int32_t* a, uint32_t* out) {
    Fq x;
    memcpy(x.d, a, sizeof x.d);
    fq_to_u32x12(x, out);
}

void f30_inv(const int32_t* a, int32_t* r) {
    Fq x;
    memcpy(x.d, a, sizeof x.d);
    Fq z = fq_inv(x);
    memcpy(r, z.d, sizeof z.d);
}
// acc (X, Y, ZZ, ZZZ as 4 x 13 digits) += (px, py), negated when neg; complete group law
void f30_madd(int32_t* acc, const int32_t* px, const int32_t* py, int neg) {
    XYZZ30 a;
    memcpy(a.X.d, acc, 52);
    memcpy(a.Y.d, acc + 13, 52);
    memcpy(a.ZZ.d, acc + 26, 52);
    memcpy(a.ZZZ.d, acc + 39, 52);
    Affine30 p;
    memcpy(p.x.d, px, 52);
    memcpy(p.y.d, py, 52);
    xyzz30_madd(a, p, neg != 0);
    memcpy(acc, a.X.d, 52);
    memcpy(acc + 13, a.Y.d, 52);
    memcpy(acc + 26, a.ZZ.d, 52);
    memcpy(acc + 39, a.ZZZ.d, 52);
}

// batched affine pair sums exactly as the accumulation kernel runs them: n pairs (a_i, b_i) with signs, forward pass
// (classification + prefix products), ONE inversion, backward pass.  out: n x (kind, x3[13], y3[13]).
void f30_pair_batch(const int32_t* ax, const int32_t* ay, const int32_t* bx, const int32_t* by, const int* nega, const int* negb,
                    int n, int32_t* out) {
    Fq* prefix = new Fq[n];
    uint32_t* kind = new uint32_t[n];
    Fq run = fq_one();
    auto pt = [](const int32_t* x, const int32_t* y, int i) {
        Affine30 p;
        memcpy(p.x.d, x + 13 * i, 52);
        memcpy(p.y.d, y + 13 * i, 52);
        return p;
    };
    for (int i = 0; i < n; i++) {
        Fq den;
        kind[i] = pair_classify(pt(ax, ay, i), nega[i], pt(bx, by, i), negb[i], den);
        prefix[i] = run;
        if (kind[i] == kPairAdd || kind[i] == kPairDouble) run = fq_mul(run, den);
    }
    Fq inv = fq_inv(run);
    for (int i = n - 1; i >= 0; i--) {
        int32_t* o = out + 27 * i;
        o[0] = (int32_t)kind[i];
        memset(o + 1, 0, 26 * 4);
        if (kind[i] != kPairAdd && kind[i] != kPairDouble) continue;
        Fq den;
        const Affine30 a = pt(ax, ay, i), b = pt(bx, by, i);
        (void)pair_classify(a, nega[i], b, negb[i], den);
        const Fq inv_den = fq_mul(inv, prefix[i]);
        inv = fq_mul(inv, den);
        const Affine30 s = pair_sum(kind[i], a, nega[i], b, negb[i], inv_den);
        memcpy(o + 1, s.x.d, 52);
        memcpy(o + 14, s.y.d, 52);
    }
    delete[] prefix;
    delete[] kind;
}

// acc (4 x 13 digits) += b (4 x 13 digits): the general addition of the tree kernels
void f30_add(int32_t* acc, const int32_t* b) {
    XYZZ30 a, o;
    memcpy(a.X.d, acc, 52); memcpy(a.Y.d, acc + 13, 52); memcpy(a.ZZ.d, acc + 26, 52); memcpy(a.ZZZ.d, acc + 39, 52);
    memcpy(o.X.d, b, 52); memcpy(o.Y.d, b + 13, 52); memcpy(o.ZZ.d, b + 26, 52); memcpy(o.ZZZ.d, b + 39, 52);
    xyzz30_add(a, o);
    memcpy(acc, a.X.d, 52); memcpy(acc + 13, a.Y.d, 52); memcpy(acc + 26, a.ZZ.d, 52); memcpy(acc + 39, a.ZZZ.d, 52);
}

// the multiplier's column sums in exact arithmetic (__int128): returns the largest |column| / 2^48 seen
int64_t f30_mul_max_column(const int32_t* a, const int32_t* b) {
    int32_t m[13];
    __int128 acc = 0, worst = 0;
    auto track = [&](__int128 v) {
        if (v < 0) v = -v;
        if (v > worst) worst = v;
    };
    for (int k = 0; k < 13; k++) {
        // worst case inside the column: sum of magnitudes
        __int128 mag = acc < 0 ? -acc : acc;
        for (int i = 0; i <= k; i++) {
            __int128 t = (__int128)a[i] * b[k - i];
            acc += t;
            mag += t < 0 ? -t : t;
        }
        for (int j = 0; j < k; j++) {
            __int128 t = (__int128)m[j] * fq_pd(k - j);
            acc += t;
            mag += t < 0 ? -t : t;
        }
        m[k] = fq_sext30((uint32_t)(uint64_t)acc * kQN0);
        __int128 t = (__int128)m[k] * fq_pd(0);
        acc += t;
        mag += t < 0 ? -t : t;
        track(mag);
        acc >>= 30;
    }
    for (int k = 13; k < 25; k++) {
        __int128 mag = acc < 0 ? -acc : acc;
        for (int i = k - 12; i < 13; i++) {
            __int128 t = (__int128)a[i] * b[k - i];
            acc += t;
            mag += t < 0 ? -t : t;
        }
        for (int j = k - 12; j < 13; j++) {
            __int128 t = (__int128)m[j] * fq_pd(k - j);
            acc += t;
            mag += t < 0 ? -t : t;
        }
        track(mag);
        int32_t dgt = fq_sext30((uint32_t)(uint64_t)acc);
        acc = (acc - dgt) >> 30;
    }
    return (int64_t)(worst >> 48);
}

// ---- the library's host field (csrc/host_field.hpp: 6 x u64, Montgomery R = 2^384) ----
void hf_mul(const uint64_t* a, const uint64_t* b, uint64_t* r) {
    kzg_host::Fp x, y;
    memcpy(x.l, a, 48);
    memcpy(y.l, b, 48);
    const kzg_host::Fp z = x * y;
    memcpy(r, z.l, 48);
}
void hf_mul_portable(const uint64_t* a, const uint64_t* b, uint64_t* r) {
    kzg_host::Fp x, y;
    memcpy(x.l, a, 48);
    memcpy(y.l, b, 48);
    const kzg_host::Fp z = kzg_host::mul_portable(x, y);
    memcpy(r, z.l, 48);
}
void hf_mul_wide(const uint64_t* a, const uint64_t* b, uint64_t* r) {
    kzg_host::Fp x, y;
    memcpy(x.l, a, 48);
    memcpy(y.l, b, 48);
    const kzg_host::Fp z = kzg_host::mul_wide(x, y);
    memcpy(r, z.l, 48);
}
void hf_inv(const uint64_t* a, uint64_t* r, uint64_t* r_fermat) {
    kzg_host::Fp x;
    memcpy(x.l, a, 48);
    const kzg_host::Fp z = kzg_host::inv(x), w = kzg_host::inv_fermat(x);
    memcpy(r, z.l, 48);
    memcpy(r_fermat, w.l, 48);
}
void hf_from_digits30(const int32_t* d, uint64_t* r) {
    const kzg_host::Fp z = kzg_host::fp_from_digits30(d);
    memcpy(r, z.l, 48);
}

// host tail in XYZZ coordinates (host_field.hpp: px_*): records are 64 x int32 as the device writes them (coordinate c at
// words 16 c).  out = affine Montgomery x (6 words), y (6 words), flag 1 = infinity
static void put_p1(const kzg_host::P1& r, uint64_t* out) {
    memcpy(out, r.x.l, 48);
    memcpy(out + 6, r.y.l, 48);
    out[12] = r.is_inf() ? 1 : 0;
}
void hf_px_sum(const int32_t* rec_a, const int32_t* rec_b, uint64_t* out_add, uint64_t* out_dbl, uint64_t* out_wsum3) {
    const kzg_host::PX a = kzg_host::px_from_record((const uint64_t*)rec_a), b = kzg_host::px_from_record((const uint64_t*)rec_b);
    put_p1(kzg_host::px_normalize(kzg_host::px_add(a, b)), out_add);
    put_p1(kzg_host::px_normalize(kzg_host::px_double(a)), out_dbl);
    // a + 2b + 3(a + b) through repeated additions of already added values (denominators far from 1)
    kzg_host::PX s = kzg_host::px_add(a, b), t = kzg_host::px_add(kzg_host::px_add(s, s), s);
    t = kzg_host::px_add(t, kzg_host::px_add(a, kzg_host::px_double(b)));
    put_p1(kzg_host::px_normalize(t), out_wsum3);
}
}
