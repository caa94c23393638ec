// srs_kernels.hip -- everything that happens once per SRS: ingest, window tables, trusted setup.
//
//  * ingest: the reference's SRS entries are blst_p1 Jacobian points with arbitrary Z
//    (src/trusted_setup.rs:54-62); the accumulation kernel wants affine points (mixed addition),
//    so they are normalised here with a per-lane batched inversion (Montgomery's trick).
//  * window tables: T[j][i] = 2^(c j) SRS[i].  With them every window of the signed-digit recoding
//    lands in ONE set of 2^(c-1) buckets, so a commitment pays the bucket reduction once instead of
//    once per window and needs no window-combining doublings.  HBM is 288 GB; the table for 2^20
//    points and 14 windows is 1.3 GiB.
//  * trusted setup on the device (reference src/trusted_setup.rs:40-62, G1 side): SRS[i] = [s^i]G1 by
//    fixed-base windowing over a table of d * 2^(8w) * G1.
#define KZG_MUL_CALL 1
#include "engine.h"
#include "g1.hip.h"

namespace kzg {

constexpr int kNormK = 32;  // points per lane in the batched inversion

KZG_DEV Fp load_fp(const uint4* __restrict__ p) {
    uint4 a = p[0], b = p[1], c = p[2];
    Fp r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    return r;
}
KZG_DEV void store_fp(uint4* __restrict__ p, const Fp& a) {
    p[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
    p[1] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
    p[2] = make_uint4(a.l[8], a.l[9], a.l[10], a.l[11]);
}
KZG_DEV Fp fp_inv_call(const Fp& a) {  // a^(p-2) on the shared multiplier
    constexpr u32 E[12] = {0xffffaaa9u, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                           0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    Fp acc = Fp::one();
#pragma unroll 1
    for (int w = 11; w >= 0; w--) {
        u32 e = E[w];
#pragma unroll 1
        for (int b = 31; b >= 0; b--) {
            acc = fsqr(acc);
            if ((e >> b) & 1) acc = fmul(acc, a);
        }
    }
    return acc;
}

// MODE 0: records are blst_p1 Jacobian {X, Y, Z} (144 B): x = X / Z^2, y = Y / Z^3
// MODE 1: records are XYZZ {X, Y, ZZ, ZZZ} (192 B):       x = X / ZZ,  y = Y / ZZZ
template <int MODE>
__global__ void __launch_bounds__(64) k_normalize(const uint4* __restrict__ in, uint32_t n, uint4* __restrict__ out,
                                                  uint4* __restrict__ prefix) {
    constexpr int REC = MODE == 0 ? 9 : 12;      // uint4 per record
    constexpr int DEN = MODE == 0 ? 6 : 9;       // uint4 offset of the denominator (Z or ZZZ)
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lo = t * kNormK;
    if (lo >= n) return;
    uint32_t hi = lo + kNormK < n ? lo + kNormK : n;
    Fp run = Fp::one();
    for (uint32_t i = lo; i < hi; i++) {
        Fp d = load_fp(in + (size_t)i * REC + DEN);
        if (!fzero(d)) run = fmul(run, d);
        store_fp(prefix + (size_t)i * 3, run);
    }
    Fp inv = fp_inv_call(run);
    for (uint32_t i = hi; i-- > lo;) {
        const uint4* rec = in + (size_t)i * REC;
        Fp d = load_fp(rec + DEN);
        uint4* o = out + (size_t)i * kAffineU4;
        if (fzero(d)) {  // infinity -> (0, 0)
            Fp zero = Fp::zero();
            store_fp(o, zero);
            store_fp(o + 3, zero);
            continue;
        }
        Fp prev = i == lo ? Fp::one() : load_fp(prefix + (size_t)(i - 1) * 3);
        Fp di = fmul(inv, prev);  // 1 / d
        inv = fmul(inv, d);
        Fp X = load_fp(rec), Y = load_fp(rec + 3);
        Fp x, y;
        if (MODE == 0) {
            Fp di2 = fsqr(di);
            x = fmul(X, di2);
            y = fmul(Y, fmul(di2, di));
        } else {
            Fp ZZ = load_fp(rec + 6);
            Fp iz = fmul(ZZ, di);  // ZZ / ZZZ = 1 / z
            x = fmul(X, fsqr(iz));
            y = fmul(Y, di);
        }
        store_fp(o, x);
        store_fp(o + 3, y);
    }
}

KZG_DEV Affine load_affine(const uint4* __restrict__ p) {
    Affine a;
    a.x = load_fp(p);
    a.y = load_fp(p + 3);
    return a;
}
KZG_DEV void store_xyzz(uint4* __restrict__ o, const XYZZ& a) {
    store_fp(o, a.X);
    store_fp(o + 3, a.Y);
    store_fp(o + 6, a.ZZ);
    store_fp(o + 9, a.ZZZ);
}

__global__ void __launch_bounds__(64) k_window_double(const uint4* __restrict__ prev, uint32_t n, uint32_t c,
                                                      uint4* __restrict__ out_xyzz) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ a = xyzz_from_affine(load_affine(prev + (size_t)i * kAffineU4));
#pragma unroll 1
    for (uint32_t k = 0; k < c; k++) a = xyzz_dbl(a);
    store_xyzz(out_xyzz + (size_t)i * 12, a);
}

// ---- fixed-base trusted setup -----------------------------------------------------------------
constexpr int kGWindows = 32;   // 8-bit windows over a 256-bit scalar
constexpr int kGDigits = 255;   // digits 1..255
size_t srs_gtable_bytes() { return (size_t)kGWindows * kGDigits * kAffineBytes; }

KZG_DEV Affine g1_generator() {
    constexpr u32 GX[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                            0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
    constexpr u32 GY[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                            0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
    Affine g;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        g.x.l[i] = GX[i];
        g.y.l[i] = GY[i];
    }
    return g;
}

// entry (w, d-1) = d * 2^(8w) * G, as XYZZ (normalised afterwards)
__global__ void __launch_bounds__(64) k_gtable_build(uint4* __restrict__ out_xyzz) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint32_t)(kGWindows * kGDigits)) return;
    uint32_t w = e / kGDigits, d = e % kGDigits + 1;
    Affine g = g1_generator();
    XYZZ acc = XYZZ::inf();
#pragma unroll 1
    for (int b = 7; b >= 0; b--) {
        acc = xyzz_dbl(acc);
        if ((d >> b) & 1) xyzz_madd(acc, g, false);
    }
#pragma unroll 1
    for (uint32_t k = 0; k < 8 * w; k++) acc = xyzz_dbl(acc);
    store_xyzz(out_xyzz + (size_t)e * 12, acc);
}

struct FrArg8 {
    uint32_t l[8];
};

__global__ void __launch_bounds__(64) k_srs_points(FrArg8 secret, uint64_t first, uint32_t n,
                                                   const uint4* __restrict__ gtable, uint4* __restrict__ out_xyzz) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s;
#pragma unroll
    for (int t = 0; t < 8; t++) s.l[t] = secret.l[t];
    s = fe_to_mont(s);  // the secret arrives as the raw 256-bit integer; this also reduces it mod r
    // s^(first + i) by square-and-multiply (reference src/trusted_setup.rs:50 walks it sequentially)
    uint64_t e = first + i;
    Fr pw = Fr::one();
    Fr base = s;
    while (e) {
        if (e & 1) pw = fe_mul(pw, base);
        base = fe_mul(base, base);
        e >>= 1;
    }
    Fr k = fe_from_mont(pw);  // canonical scalar, as Scalar::to_le_bytes (src/scalar.rs:83-93)
    XYZZ acc = XYZZ::inf();
#pragma unroll 1
    for (int w = 0; w < kGWindows; w++) {
        u32 d = k.l[0] & 0xffu;
#pragma unroll
        for (int t = 0; t < 7; t++) k.l[t] = (k.l[t] >> 8) | (k.l[t + 1] << 24);
        k.l[7] >>= 8;
        if (d) {
            Affine p = load_affine(gtable + (size_t)(w * kGDigits + (d - 1)) * kAffineU4);
            xyzz_madd(acc, p, false);
        }
    }
    store_xyzz(out_xyzz + (size_t)i * 12, acc);
}

void launch_jacobian_to_affine(hipStream_t s, const void* d_jac, uint32_t n, void* d_out, void* d_prefix) {
    if (!n) return;
    uint32_t lanes = (n + kNormK - 1) / kNormK;
    hipLaunchKernelGGL(k_normalize<0>, dim3((lanes + 63) / 64), dim3(64), 0, s, (const uint4*)d_jac, n, (uint4*)d_out,
                       (uint4*)d_prefix);
}
static void launch_xyzz_to_affine(hipStream_t s, const void* d_xyzz, uint32_t n, void* d_out, void* d_prefix) {
    if (!n) return;
    uint32_t lanes = (n + kNormK - 1) / kNormK;
    hipLaunchKernelGGL(k_normalize<1>, dim3((lanes + 63) / 64), dim3(64), 0, s, (const uint4*)d_xyzz, n, (uint4*)d_out,
                       (uint4*)d_prefix);
}
void launch_table_window(hipStream_t s, const void* d_prev, uint32_t n, uint32_t c, void* d_xyzz_tmp, void* d_prefix,
                         void* d_next) {
    if (!n) return;
    hipLaunchKernelGGL(k_window_double, dim3((n + 63) / 64), dim3(64), 0, s, (const uint4*)d_prev, n, c,
                       (uint4*)d_xyzz_tmp);
    launch_xyzz_to_affine(s, d_xyzz_tmp, n, d_next, d_prefix);
}
void launch_srs_generate(hipStream_t s, const uint32_t* secret_raw8, uint64_t first, uint32_t n, void* d_gtable,
                         void* d_xyzz_tmp, void* d_prefix, void* d_out) {
    if (!n) return;
    const uint32_t ge = kGWindows * kGDigits;
    // the temporaries are sized for max(n, ge) records by the caller
    hipLaunchKernelGGL(k_gtable_build, dim3((ge + 63) / 64), dim3(64), 0, s, (uint4*)d_xyzz_tmp);
    launch_xyzz_to_affine(s, d_xyzz_tmp, ge, d_gtable, d_prefix);
    FrArg8 sec;
    for (int i = 0; i < 8; i++) sec.l[i] = secret_raw8[i];
    hipLaunchKernelGGL(k_srs_points, dim3((n + 63) / 64), dim3(64), 0, s, sec, first, n, (const uint4*)d_gtable,
                       (uint4*)d_xyzz_tmp);
    launch_xyzz_to_affine(s, d_xyzz_tmp, n, d_out, d_prefix);
}
}  // namespace kzg
