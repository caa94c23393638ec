// microbench.hip -- gfx950 integer-VALU ceilings for the field arithmetic (not part of the library).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
// Prints one JSON object per measurement.  The v_mad_u64_u32 issue rate measured here is the VALU
// peak the MSM kernels are quoted against ("must be measured on the box").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "legacy_field/g1.hip.h"  // rounds 1-2: the 12 x u32 field, kept for the comparisons
#include "../kzg_poly_commit_exploration_amd/csrc/g1_30.hip.h"

using namespace kzg;

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int CHAINS = 8;

__global__ void __launch_bounds__(256) k_mad(u64* out, int iters) {
    u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 77u;
    u64 acc[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; k++) acc[k] = a + k;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < CHAINS; k++) acc[k] = mad64((u32)acc[k] ^ a, b + k, acc[k] >> 32);
        }
    }
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// how many full-rate VALU instructions hide behind one quarter-rate v_mad_u64_u32?  R independent v_add_u32
// per multiply-add, 8 independent multiply-add chains.
template <int R>
__global__ void __launch_bounds__(256) k_mad_mix(u64* out, int iters) {
    u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 77u;
    u64 acc[CHAINS];
    u32 x[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; k++) {
        acc[k] = a + k;
        x[k] = b + k;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < CHAINS; k++) {
                asm volatile("v_mad_u64_u32 %0, s[72:73], %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b) : "s72", "s73");
#pragma unroll
                for (int j = 0; j < R; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[(k + j + 1) & 7]) : "v"(a));
            }
        }
    }
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; k++) s ^= acc[k] + x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// carry handling of six multiply-adds: (A) one v_addc per carry mask, as the multiplier does today; (B) the six masks
// compressed by a carry-save adder tree on the SCALAR unit (17 s_*_b64) into masks of weight 1, 2, 4 -> three v_addc.
template <int MODE>
__global__ void __launch_bounds__(256) k_mad_carry(u64* out, int iters) {
    u32 a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 77u;
    u64 acc = a;
    u32 o1 = 0, o2 = 0, o4 = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (MODE == 0) {
                asm volatile(
                    "v_mad_u64_u32 %0, s[72:73], %2, %3, %0\n\tv_mad_u64_u32 %0, s[74:75], %2, %3, %0\n\t"
                    "v_mad_u64_u32 %0, s[76:77], %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, s[72:73]\n\t"
                    "v_mad_u64_u32 %0, s[72:73], %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, s[74:75]\n\t"
                    "v_mad_u64_u32 %0, s[74:75], %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, s[76:77]\n\t"
                    "v_mad_u64_u32 %0, s[76:77], %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, s[72:73]\n\t"
                    "v_addc_co_u32 %1, vcc, 0, %1, s[74:75]\n\tv_addc_co_u32 %1, vcc, 0, %1, s[76:77]"
                    : "+v"(acc), "+v"(o1)
                    : "v"(a), "v"(b)
                    : "vcc", "s72", "s73", "s74", "s75", "s76", "s77");
            } else {
                asm volatile(
                    "v_mad_u64_u32 %0, s[72:73], %4, %5, %0\n\tv_mad_u64_u32 %0, s[74:75], %4, %5, %0\n\t"
                    "v_mad_u64_u32 %0, s[76:77], %4, %5, %0\n\tv_mad_u64_u32 %0, s[78:79], %4, %5, %0\n\t"
                    "v_mad_u64_u32 %0, s[80:81], %4, %5, %0\n\tv_mad_u64_u32 %0, s[82:83], %4, %5, %0\n\t"
                    "s_xor_b64 s[84:85], s[72:73], s[74:75]\n\ts_and_b64 s[86:87], s[72:73], s[74:75]\n\t"
                    "s_and_b64 s[88:89], s[76:77], s[84:85]\n\ts_xor_b64 s[84:85], s[84:85], s[76:77]\n\t"
                    "s_or_b64 s[86:87], s[86:87], s[88:89]\n\t"                       // s1 = 84, c1 = 86
                    "s_xor_b64 s[88:89], s[78:79], s[80:81]\n\ts_and_b64 s[90:91], s[78:79], s[80:81]\n\t"
                    "s_and_b64 s[72:73], s[82:83], s[88:89]\n\ts_xor_b64 s[88:89], s[88:89], s[82:83]\n\t"
                    "s_or_b64 s[90:91], s[90:91], s[72:73]\n\t"                       // s2 = 88, c2 = 90
                    "s_and_b64 s[72:73], s[84:85], s[88:89]\n\ts_xor_b64 s[84:85], s[84:85], s[88:89]\n\t"  // c3 = 72, s = 84
                    "s_xor_b64 s[74:75], s[86:87], s[90:91]\n\ts_and_b64 s[76:77], s[86:87], s[90:91]\n\t"
                    "s_and_b64 s[78:79], s[72:73], s[74:75]\n\ts_xor_b64 s[74:75], s[74:75], s[72:73]\n\t"  // t = 74
                    "s_or_b64 s[76:77], s[76:77], s[78:79]\n\t"                        // u = 76
                    "v_addc_co_u32 %1, vcc, 0, %1, s[84:85]\n\tv_addc_co_u32 %2, vcc, 0, %2, s[74:75]\n\t"
                    "v_addc_co_u32 %3, vcc, 0, %3, s[76:77]"
                    : "+v"(acc), "+v"(o1), "+v"(o2), "+v"(o4)
                    : "v"(a), "v"(b)
                    : "vcc", "scc", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84",
                      "s85", "s86", "s87", "s88", "s89", "s90", "s91");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + o1 + 2 * o2 + 4 * o4;
}

__global__ void __launch_bounds__(256) k_mullo(u64* out, int iters) {
    u32 a = threadIdx.x * 2654435761u + 12345u;
    u32 acc[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; k++) acc[k] = a + k;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < CHAINS; k++) acc[k] = acc[k] * (a | 1u);
        }
    }
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mulhi(u64* out, int iters) {
    u32 a = threadIdx.x * 2654435761u + 12345u;
    u32 acc[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; k++) acc[k] = a + k * 0x9e3779b9u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < CHAINS; k++) acc[k] = __umulhi(acc[k] | 0x80000000u, a | 0xC0000000u) + k;
        }
    }
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mad24(u64* out, int iters) {
    u32 a = (threadIdx.x * 2654435761u + 12345u) & 0xffffffu;
    u32 acc[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; k++) acc[k] = a + k;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < CHAINS; k++) acc[k] = __umul24(acc[k], a) + (acc[k] >> 3);
        }
    }
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_addc(u64* out, int iters) {
    u32 a = threadIdx.x * 2654435761u + 12345u;
    u32 acc[12], b[12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
        acc[k] = a + k;
        b[k] = a * (k + 3);
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            u32 c = 0;
#pragma unroll
            for (int k = 0; k < 12; k++) acc[k] = addc(acc[k], b[k], c);
            b[0] += c;
        }
    }
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_dfma(u64* out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999;
    double acc[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; k++) acc[k] = a + k;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < CHAINS; k++) acc[k] = __fma_rn(acc[k], b, a);
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u64)s;
}

__global__ void __launch_bounds__(256) k_fpmul(u32* io, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x, y;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        x.l[i] = io[i];
        y.l[i] = io[12 + i];
    }
    x.l[0] ^= (tid & 0xff);  // still < p (top limb untouched)
    for (int it = 0; it < iters; it++) {
        x = fe_mul(x, y);
        y = fe_mul(y, x);
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            io[24 + i] = x.l[i];
            io[36 + i] = y.l[i];
        }
    }
    if (x.l[3] == 0x12345 && y.l[2] == 77) io[48] = 1;  // keep live for all threads
}

__global__ void __launch_bounds__(256) k_fpmul_fips(u32* io, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x, y;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        x.l[i] = io[i];
        y.l[i] = io[12 + i];
    }
    x.l[0] ^= (tid & 0xff);
    for (int it = 0; it < iters; it++) {
        x = fe_mul_fips(x, y);
        y = fe_mul_fips(y, x);
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            io[24 + i] = x.l[i];
            io[36 + i] = y.l[i];
        }
    }
    if (x.l[3] == 0x12345 && y.l[2] == 77) io[48] = 1;
}


// ---- signed radix-2^30 field (field30.hip.h): the same three loops as k_fpmul_fips / k_madd ----------------------
__global__ void __launch_bounds__(256) k_fqmul(u32* io, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fq x = fq_from_u32x12(io), y = fq_from_u32x12(io + 12);
    x.d[0] ^= (tid & 0xff);
    x = fq_mul(x, fq_one());
    y = fq_mul(y, fq_one());
    for (int it = 0; it < iters; it++) {
        x = fq_mul(x, y);
        y = fq_mul(y, x);
    }
    if (tid == 0) {
        fq_to_u32x12(x, io + 24);
        fq_to_u32x12(y, io + 36);
    }
    if (x.d[3] == 0x12345 && y.d[2] == 77) io[48] = 1;
}
__global__ void __launch_bounds__(256) k_fqsqr(u32* io, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fq x = fq_from_u32x12(io);
    x.d[0] ^= (tid & 0xff);
    x = fq_mul(x, fq_one());
    for (int it = 0; it < iters; it++) x = fq_sqr(x);
    if (tid == 0) fq_to_u32x12(x, io + 24);
    if (x.d[3] == 0x12345) io[48] = 1;
}
__global__ void __launch_bounds__(256) k_fpsqr_fips(u32* io, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp x;
#pragma unroll
    for (int i = 0; i < 12; i++) x.l[i] = io[i];
    x.l[0] ^= (tid & 0xff);
    for (int it = 0; it < iters; it++) x = fe_sqr_fips<FpParams, true>(x);
    if (tid == 0) {
        Fp c = fp_canon(x);
#pragma unroll
        for (int i = 0; i < 12; i++) io[24 + i] = c.l[i];
    }
    if (x.l[3] == 0x12345) io[48] = 1;
}
__global__ void __launch_bounds__(256) k_madd30(u32* io, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Affine30 p;
    p.x = fq_mul(fq_from_u32x12(io), fq_one());
    p.y = fq_mul(fq_from_u32x12(io + 12), fq_one());
    XYZZ30 acc = xyzz30_inf();
    xyzz30_madd(acc, p, false);
    xyzz30_madd(acc, p, false);  // 2P through the doubling branch
    xyzz30_dbl_inplace(acc);     // 4P
    for (int it = 0; it < iters; it++) xyzz30_madd(acc, p, (it + tid) & 1);
    if (tid == 0) {
        fq_to_u32x12(acc.X, io + 24);
        fq_to_u32x12(acc.Y, io + 36);
        fq_to_u32x12(acc.ZZ, io + 48);
        fq_to_u32x12(acc.ZZZ, io + 60);
    }
    if (acc.X.d[3] == 0x12345 && acc.Y.d[2] == 77) io[80] = 1;
}

// ---- quad-cooperative general addition (xyzz30_add_quad) against the one-lane xyzz30_add -----------------------
// Every quad builds A = (4 + tid/4) P and B = 3P, adds them both ways; quad 0 writes both results and lane q of quad 0
// writes the stage-1 product it computed (words 128 + 13 q).
__global__ void __launch_bounds__(256) k_addquad(u32* io, int iters, int* cmp) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t q = threadIdx.x & 3u;
    Affine30 p;
    p.x = fq_mul(fq_from_u32x12(io), fq_one());
    p.y = fq_mul(fq_from_u32x12(io + 12), fq_one());
    XYZZ30 a = xyzz30_inf(), b = xyzz30_inf();
    xyzz30_madd(a, p, false);
    xyzz30_madd(a, p, false);
    xyzz30_dbl_inplace(a);  // 4P
    for (int k = 0; k < (tid / 4) % 5; k++) xyzz30_madd(a, p, false);
    xyzz30_madd(b, p, false);
    xyzz30_madd(b, p, false);
    xyzz30_madd(b, p, false);  // 3P
    XYZZ30 r1 = a, r2 = a;
    for (int it = 0; it < iters; it++) {
        xyzz30_add(r1, b);
        xyzz30_add_quad(r2, b, q);
    }
    // projective equality r1 == r2: X1 ZZ2 == X2 ZZ1, Y1 ZZZ2 == Y2 ZZZ1
    const bool same = fq_is_zero(fq_norm(fq_sub_raw(fq_mul(r1.X, r2.ZZ), fq_mul(r2.X, r1.ZZ)))) &&
                      fq_is_zero(fq_norm(fq_sub_raw(fq_mul(r1.Y, r2.ZZZ), fq_mul(r2.Y, r1.ZZZ))));
    if (!same) atomicAdd(cmp, 1);
    bool digits = true;
    for (int i = 0; i < kQ; i++) digits = digits && r1.X.d[i] == r2.X.d[i] && r1.Y.d[i] == r2.Y.d[i] && r1.ZZ.d[i] == r2.ZZ.d[i] && r1.ZZZ.d[i] == r2.ZZZ.d[i];
    if (!digits) atomicAdd(cmp + 1, 1);
    if (tid < 16) {
        int m = 0;
        bool e;
        e = true; for (int i = 0; i < kQ; i++) e = e && r1.X.d[i] == r2.X.d[i]; m |= e ? 1 : 0;
        e = true; for (int i = 0; i < kQ; i++) e = e && r1.Y.d[i] == r2.Y.d[i]; m |= e ? 2 : 0;
        e = true; for (int i = 0; i < kQ; i++) e = e && r1.ZZ.d[i] == r2.ZZ.d[i]; m |= e ? 4 : 0;
        e = true; for (int i = 0; i < kQ; i++) e = e && r1.ZZZ.d[i] == r2.ZZZ.d[i]; m |= e ? 8 : 0;
        cmp[8 + tid] = m;
        cmp[32 + tid] = r2.Y.d[0];
        cmp[48 + tid] = r2.Y.d[12];
        cmp[64 + tid] = r1.Y.d[0];
        cmp[80 + tid] = r1.Y.d[12];
    }
    if (tid == 0) {
        fq_to_u32x12(r1.X, io + 24);
        fq_to_u32x12(r1.Y, io + 36);
        fq_to_u32x12(r2.X, io + 48);
        fq_to_u32x12(r2.Y, io + 60);
    }
}

// ---- one workgroup of 64 quads folding 64 points through LDS (the shape of k_tree_sum / tree64), with the constant
// 100 MHz clock read by thread 0 after every level: where does a tree level's time go?
__global__ void __launch_bounds__(256, 1) k_tree_probe(u32* io, unsigned long long* stamps, int mode) {
    // mode 0: one tree over the 64 quads; mode 2 / 3 / 4: independent trees over groups of 16 / 8 / 4 quads (every wave
    // keeps 1, 2 or 4 quads busy down to the last level)
    __shared__ uint32_t lds[4 * kQ * 64];
    const int t = threadIdx.x / 4;
    const uint32_t q = threadIdx.x & 3u;
    const bool lead = q == 0;
    const int gsz = mode == 0 ? 64 : (mode == 2 || mode == 5) ? 16 : mode == 3 ? 8 : 4;
    const int l = t & (gsz - 1);
    Affine30 p;
    p.x = fq_mul(fq_from_u32x12(io), fq_one());
    p.y = fq_mul(fq_from_u32x12(io + 12), fq_one());
    XYZZ30 acc = xyzz30_inf();
    xyzz30_madd(acc, p, false);
    xyzz30_madd(acc, p, false);
    for (int k = 0; k < t; k++) xyzz30_madd(acc, p, false);  // (2 + t) P: all 64 differ, and so do all partial sums
    __syncthreads();
    stamps += blockIdx.x * 8;
    if (threadIdx.x == 0) stamps[0] = wall_clock64();
    int lvl = 1;
    for (int off = gsz >> 1; off >= 1; off >>= 1, lvl++) {
        __syncthreads();
        if (lead && l >= off && l < 2 * off) {
            const Fq* f[4] = {&acc.X, &acc.Y, &acc.ZZ, &acc.ZZZ};
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int i = 0; i < kQ; i++) lds[(c * kQ + i) * 64 + (t - off)] = (uint32_t)f[c]->d[i];
        }
        __syncthreads();
        if (mode == 5) {  // every quad enters; those without a partner bring an operand at infinity
            XYZZ30 o = xyzz30_inf();
            if (l < off) {
                Fq* f[4] = {&o.X, &o.Y, &o.ZZ, &o.ZZZ};
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int i = 0; i < kQ; i++) f[c]->d[i] = (int32_t)lds[(c * kQ + i) * 64 + t];
            }
            xyzz30_add_quad_dense(acc, o, q);
        } else if (l < off) {
            XYZZ30 o;
            Fq* f[4] = {&o.X, &o.Y, &o.ZZ, &o.ZZZ};
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int i = 0; i < kQ; i++) f[c]->d[i] = (int32_t)lds[(c * kQ + i) * 64 + t];
            xyzz30_add_quad(acc, o, q);
        }
        if (threadIdx.x == 0) stamps[lvl] = wall_clock64();
    }
    for (; lvl < 7; lvl++)
        if (threadIdx.x == 0) stamps[lvl] = stamps[lvl - 1];
    if (threadIdx.x == 0) {
        fq_to_u32x12(acc.X, io + 24);
        fq_to_u32x12(acc.Y, io + 36);
    }
}
// Four waves of one workgroup running the cooperative addition out of phase with each other (wave w starts w * skew
// later): does a wave's addition take longer when its neighbours on the CU are at other places of the same code?
__global__ void __launch_bounds__(256, 1) k_phase_probe(u32* io, unsigned long long* stamps, int skew_sleeps, int active_waves) {
    const uint32_t q = threadIdx.x & 3u;
    const int w = threadIdx.x >> 6;
    Affine30 p;
    p.x = fq_mul(fq_from_u32x12(io), fq_one());
    p.y = fq_mul(fq_from_u32x12(io + 12), fq_one());
    XYZZ30 a = xyzz30_inf(), b = xyzz30_inf();
    xyzz30_madd(a, p, false);
    xyzz30_madd(a, p, false);
    xyzz30_dbl_inplace(a);
    xyzz30_madd(b, p, false);
    xyzz30_madd(b, p, false);
    xyzz30_madd(b, p, false);
    __syncthreads();
    if (w < active_waves) {
        for (int k = 0; k < w * skew_sleeps; k++) __builtin_amdgcn_s_sleep(16);
        for (int it = 0; it < 4; it++) {
            const unsigned long long t0 = wall_clock64();
            xyzz30_add_quad(a, b, q);
            const unsigned long long t1 = wall_clock64();
            if ((threadIdx.x & 63) == 0) {
                stamps[(blockIdx.x * 4 + w) * 8 + it * 2] = t0;
                stamps[(blockIdx.x * 4 + w) * 8 + it * 2 + 1] = t1;
            }
        }
    }
    if (a.X.d[3] == 0x12345 && a.Y.d[2] == 77) io[80] = 1;
}

// One wave: `active` of its 16 quads run the cooperative addition; the others either skip it (mode 0) or enter it
// with an operand at infinity and leave through its early exit (mode 1), as in the step loop of k_small_msm.
__global__ void __launch_bounds__(256, 1) k_partial_probe(u32* io, unsigned long long* stamps, int active, int mode) {
    const uint32_t q = threadIdx.x & 3u;
    const int quad = (threadIdx.x & 63) >> 2;
    Affine30 p;
    p.x = fq_mul(fq_from_u32x12(io), fq_one());
    p.y = fq_mul(fq_from_u32x12(io + 12), fq_one());
    XYZZ30 a = xyzz30_inf(), b = xyzz30_inf();
    xyzz30_madd(a, p, false);
    xyzz30_madd(a, p, false);
    xyzz30_dbl_inplace(a);
    xyzz30_madd(b, p, false);
    xyzz30_madd(b, p, false);
    xyzz30_madd(b, p, false);
    if (quad >= active && mode == 1) b = xyzz30_inf();
    __syncthreads();
    for (int it = 0; it < 4; it++) {
        const unsigned long long t0 = wall_clock64();
        if (quad < active || mode == 1) xyzz30_add_quad(a, b, q);
        const unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) {
            stamps[it * 2] = t0;
            stamps[it * 2 + 1] = t1;
        }
    }
    if (a.X.d[3] == 0x12345 && a.Y.d[2] == 77) io[80] = 1;
}
__global__ void __launch_bounds__(256) k_dpp_probe(int* out) {
    const int v = threadIdx.x * 10;
    out[threadIdx.x] = __builtin_amdgcn_mov_dpp(v, 2 * 0x55, 0xf, 0xf, true);
}
// instruction mix of an fp64-FMA multiplier (Emmart et al.: 52-bit limbs, 8 limbs for 381 bits): per limb product
// two v_fma_f64 (high and low half), one v_add_f64 (the correction term) and two 64-bit integer adds; 128 limb
// products per Montgomery product.  NOT a multiplier -- the dependency shape and the instruction counts only.
__global__ void __launch_bounds__(256) k_dpf_mix(u64* out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999 + blockIdx.x * 1e-12;
    const double c1 = 20282409603651670423947251286016.0, c2 = c1 + 4503599627370496.0;  // 2^104, 2^104 + 2^52
    u64 acc[4] = {1, 2, 3, 4};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 128; q++) {
            double hi = __fma_rn(a, b, c1);
            double sub = c2 - hi;
            double lo = __fma_rn(a, b, sub);
            acc[q & 3] += (u64)__double_as_longlong(hi);
            acc[(q + 1) & 3] += (u64)__double_as_longlong(lo);
            a = lo * 1e-30 + a;  // keep the chain data dependent (one more fma; counted in the note)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

__global__ void __launch_bounds__(256) k_frmul(u32* io, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fr x, y;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        x.l[i] = io[i];
        y.l[i] = io[8 + i];
    }
    x.l[0] ^= (tid & 0xff);
    for (int it = 0; it < iters; it++) {
        x = fe_mul(x, y);
        y = fe_mul(y, x);
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            io[16 + i] = x.l[i];
            io[24 + i] = y.l[i];
        }
    }
    if (x.l[3] == 0x12345 && y.l[2] == 77) io[48] = 1;
}

__global__ void __launch_bounds__(256) k_madd(u32* io, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Affine p;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        p.x.l[i] = io[i];
        p.y.l[i] = io[12 + i];
    }
    XYZZ acc = XYZZ::inf();
    Affine q = p;
    xyzz_madd(acc, p, false);
    xyzz_madd(acc, p, false);  // acc = 2P (exercises the rare doubling path)
    acc = xyzz_dbl(acc);       // 4P: the loop below walks 4P <-> 3P/5P and never meets +-P
    for (int it = 0; it < iters; it++) {
        xyzz_madd(acc, q, (it + tid) & 1);
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            io[24 + i] = acc.X.l[i];
            io[36 + i] = acc.Y.l[i];
            io[48 + i] = acc.ZZ.l[i];
            io[60 + i] = acc.ZZZ.l[i];
        }
    }
    if (acc.X.l[3] == 0x12345 && acc.Y.l[2] == 77) io[80] = 1;
}

__global__ void __launch_bounds__(256) k_add(u32* io, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Affine p;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        p.x.l[i] = io[i];
        p.y.l[i] = io[12 + i];
    }
    XYZZ acc = xyzz_from_affine(p);
    acc = xyzz_dbl(acc);
    XYZZ run = xyzz_dbl(acc);
    xyzz_madd(run, p, tid & 1);
    for (int it = 0; it < iters; it++) {
        xyzz_add(run, acc);
        xyzz_add(acc, run);
    }
    if (acc.X.l[3] == 0x12345 && acc.Y.l[2] == 77) io[80] = 1;
}

// calibration of the rocprofv3 FETCH_SIZE counter on the accumulation kernel's access pattern:
// every lane reads `iters` random 96-byte records (6 x dwordx4) from a table larger than the
// Infinity Cache.  Requested bytes are known: lanes * iters * 96.
__global__ void __launch_bounds__(256) k_gather96(const uint4* __restrict__ table, uint32_t records, int iters,
                                                  u64* out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t state = tid * 2654435761u + 12345u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++) {
        state = state * 1664525u + 1013904223u;
        uint32_t idx = (uint32_t)(((u64)state * records) >> 32);
        const uint4* p = table + (size_t)idx * 6;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            uint4 v = p[q];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    out[tid] = (u64)acc.x ^ acc.y ^ ((u64)(acc.z ^ acc.w) << 32);
}

// same with 128-byte records and a footprint parameter: does a window table of tens of GB (one level per
// scalar bit) still gather at the rate the accumulation kernel needs (~5e9 records/s)?  TLB reach test.
__global__ void __launch_bounds__(256) k_gather128(const uint4* __restrict__ table, uint32_t records, int iters,
                                                   u64* out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t state = tid * 2654435761u + 12345u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; it++) {
        state = state * 1664525u + 1013904223u;
        uint32_t idx = (uint32_t)(((u64)(state ^ (state >> 15)) * records) >> 32);
        const uint4* p = table + (size_t)idx * 8;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            uint4 v = p[q];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    out[tid] = (u64)acc.x ^ acc.y ^ ((u64)(acc.z ^ acc.w) << 32);
}

template <class K, class... A>
static double time_kernel(K kern, int grid, int block, int reps, A... args) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, args...);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, args...);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static void hexout(const char* name, const u32* l, int n) {
    printf("\"%s\": \"", name);
    for (int i = n - 1; i >= 0; i--) printf("%08x", l[i]);
    printf("\"");
}

static int run_gather_calibration() {
    const uint32_t records = 15u * 1048577u;  // the 2^20 window table: 1.41 GiB
    uint4* table;
    u64* out;
    CHECK(hipMalloc(&table, (size_t)records * 96));
    CHECK(hipMemset(table, 0x5a, (size_t)records * 96));
    const int lanes = 262144, iters = 60;
    CHECK(hipMalloc(&out, sizeof(u64) * lanes));
    double ms = time_kernel(k_gather96, lanes / 256, 256, 3, (const uint4*)table, records, iters, out);
    double bytes = (double)lanes * iters * 96.0;
    printf("{\"bench\": \"gather96\", \"requested_bytes_per_launch\": %.0f, \"ms\": %.4f, \"GBs\": %.1f}\n", bytes, ms,
           bytes / ms / 1e6);
    return 0;
}

static int run_gather_footprints() {
    const int lanes = 196608, iters = 80;  // the accumulation kernel's shape: 15.7M gathers per launch
    u64* out;
    CHECK(hipMalloc(&out, sizeof(u64) * lanes));
    for (double gib : {2.0, 8.0, 34.0, 137.0}) {
        size_t bytes = (size_t)(gib * 1073741824.0);
        uint32_t records = (uint32_t)(bytes / 128);
        uint4* table;
        if (hipMalloc(&table, bytes) != hipSuccess) {
            printf("{\"bench\": \"gather128\", \"footprint_GiB\": %.0f, \"error\": \"hipMalloc failed\"}\n", gib);
            (void)hipGetLastError();
            continue;
        }
        CHECK(hipMemset(table, 0x5a, bytes));
        double ms = time_kernel(k_gather128, lanes / 256, 256, 3, (const uint4*)table, records, iters, out);
        printf("{\"bench\": \"gather128\", \"footprint_GiB\": %.0f, \"gathers\": %d, \"ms\": %.4f, \"Ggather_s\": %.2f}\n", gib,
               lanes * iters, ms, (double)lanes * iters / ms / 1e6);
        fflush(stdout);
        CHECK(hipFree(table));
    }
    return 0;
}

template <int R>
static void run_mix(u64* out, int cus, int wps) {
    const int iters = 2000, grid = cus * wps;
    double ms = time_kernel(k_mad_mix<R>, grid, 256, 5, out, iters);
    double n_mad = (double)grid * 256 * iters * 8 * CHAINS;
    printf("{\"bench\": \"mad_plus_adds\", \"adds_per_mad\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"Tmad_s\": %.2f}\n", R, wps, ms,
           n_mad / ms / 1e9);
}
static int run_mix_all() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    u64* out;
    CHECK(hipMalloc(&out, sizeof(u64) * 256 * 8 * 1024));
    for (int wps : {1, 2, 3, 4}) {
        run_mix<0>(out, cus, wps);
        run_mix<1>(out, cus, wps);
        run_mix<2>(out, cus, wps);
        run_mix<3>(out, cus, wps);
        run_mix<4>(out, cus, wps);
        run_mix<6>(out, cus, wps);
    }
    return 0;
}

static int run_carry_debug() {
    u64* out;
    CHECK(hipMalloc(&out, sizeof(u64) * 256 * 8));
    CHECK(hipMemset(out, 0xff, sizeof(u64) * 256 * 8));
    for (int mode = 0; mode < 2; mode++) {
        if (mode == 0) hipLaunchKernelGGL(k_mad_carry<0>, dim3(2), dim3(256), 0, 0, out, 3);
        else hipLaunchKernelGGL(k_mad_carry<1>, dim3(2), dim3(256), 0, 0, out, 3);
        hipError_t e1 = hipGetLastError();
        hipError_t e2 = hipDeviceSynchronize();
        u64 h[4];
        CHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
        printf("mode %d launch=%s sync=%s out0=%016llx out1=%016llx\n", mode, hipGetErrorString(e1), hipGetErrorString(e2),
               (unsigned long long)h[0], (unsigned long long)h[1]);
    }
    return 0;
}

static int run_carry() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    u64* out;
    CHECK(hipMalloc(&out, sizeof(u64) * 256 * 8 * 1024));
    std::vector<u64> h0(256), h1(256);
    for (int wps : {1, 2, 3, 4}) {
        const int iters = 2000, grid = cus * wps;
        double n_mad = (double)grid * 256 * iters * 8 * 6;
        double ms0 = time_kernel(k_mad_carry<0>, grid, 256, 5, out, iters);
        CHECK(hipMemcpy(h0.data(), out, 256 * 8, hipMemcpyDeviceToHost));
        double ms1 = time_kernel(k_mad_carry<1>, grid, 256, 5, out, iters);
        CHECK(hipMemcpy(h1.data(), out, 256 * 8, hipMemcpyDeviceToHost));
        bool same = h0 == h1;
        printf("{\"bench\": \"carry_handling\", \"waves_per_simd\": %d, \"addc_per_mad_ms\": %.4f, \"salu_tree_ms\": %.4f, \"Tmad_s_addc\": %.2f, \"Tmad_s_tree\": %.2f, \"same_result\": %s}\n",
               wps, ms0, ms1, n_mad / ms0 / 1e9, n_mad / ms1 / 1e9, same ? "true" : "false");
    }
    return 0;
}


static bool canon_eq(const u32* a_lazy, const u32* b) {
    // a_lazy in [0, 2p) -> canonical, compare with b (canonical)
    static const u32 P[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                              0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
    u32 d[12];
    long long br = 0;
    for (int i = 0; i < 12; i++) {
        long long v = (long long)a_lazy[i] - P[i] - br;
        d[i] = (u32)v;
        br = v < 0 ? 1 : 0;
    }
    const u32* a = br ? a_lazy : d;
    return memcmp(a, b, 48) == 0;
}

// Fp multipliers side by side: 12 x u32 FIPS (shipped in round 1), signed radix-2^30 (13 digits), fp64 instruction mix
static int run_field30() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    const u32 GX[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                        0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
    const u32 GY[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                        0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
    u32 h[128], h2[128], hin[128];
    u32* dio;
    u64* out;
    CHECK(hipMalloc(&dio, sizeof h));
    CHECK(hipMalloc(&out, sizeof(u64) * 256 * 8 * 1024));
    auto reset = [&] {
        memset(hin, 0, sizeof hin);
        memcpy(hin, GX, 48);
        memcpy(hin + 12, GY, 48);
        CHECK(hipMemcpy(dio, hin, sizeof hin, hipMemcpyHostToDevice));
    };
    // correctness first: the same short loops through both representations
    reset();
    hipLaunchKernelGGL(k_fpmul_fips, dim3(1), dim3(64), 0, 0, dio, 3);
    CHECK(hipMemcpy(h, dio, sizeof h, hipMemcpyDeviceToHost));
    reset();
    hipLaunchKernelGGL(k_fqmul, dim3(1), dim3(64), 0, 0, dio, 3);
    CHECK(hipMemcpy(h2, dio, sizeof h2, hipMemcpyDeviceToHost));
    bool ok_mul = canon_eq(h + 24, h2 + 24) && canon_eq(h + 36, h2 + 36);
    reset();
    hipLaunchKernelGGL(k_fpsqr_fips, dim3(1), dim3(64), 0, 0, dio, 5);
    CHECK(hipMemcpy(h, dio, sizeof h, hipMemcpyDeviceToHost));
    reset();
    hipLaunchKernelGGL(k_fqsqr, dim3(1), dim3(64), 0, 0, dio, 5);
    CHECK(hipMemcpy(h2, dio, sizeof h2, hipMemcpyDeviceToHost));
    bool ok_sqr = canon_eq(h + 24, h2 + 24);
    reset();
    hipLaunchKernelGGL(k_madd, dim3(1), dim3(64), 0, 0, dio, 7);
    CHECK(hipMemcpy(h, dio, sizeof h, hipMemcpyDeviceToHost));
    reset();
    hipLaunchKernelGGL(k_madd30, dim3(1), dim3(64), 0, 0, dio, 7);
    CHECK(hipMemcpy(h2, dio, sizeof h2, hipMemcpyDeviceToHost));
    bool ok_madd = canon_eq(h + 24, h2 + 24) && canon_eq(h + 36, h2 + 36) && canon_eq(h + 48, h2 + 48) && canon_eq(h + 60, h2 + 60);
    printf("{\"probe\": \"field30_vs_fips\", \"mul_equal\": %s, \"sqr_equal\": %s, \"madd_equal\": %s}\n", ok_mul ? "true" : "false",
           ok_sqr ? "true" : "false", ok_madd ? "true" : "false");
    // latency of ONE wave alone on the chip (what the finalisation / reduction trees pay per dependent addition)
    {
        reset();
        double ms = time_kernel(k_fqmul, 1, 64, 5, dio, 200);
        printf("{\"bench\": \"lone_wave_fp_mul_signed30\", \"us_per_product\": %.3f}\n", ms * 1e3 / 400.0);
        reset();
        ms = time_kernel(k_madd30, 1, 64, 5, dio, 64);
        printf("{\"bench\": \"lone_wave_xyzz_madd_signed30\", \"us_per_addition\": %.3f}\n", ms * 1e3 / 64.0);
        reset();
        ms = time_kernel(k_madd30, 256, 64, 5, dio, 64);
        printf("{\"bench\": \"one_wave_per_cu_xyzz_madd_signed30\", \"us_per_addition\": %.3f}\n", ms * 1e3 / 64.0);
    }
    for (int wps = 1; wps <= 4; wps++) {
        int grid = cus * wps;
        const int fit = 200, block = 256;
        double nmul = (double)grid * block * fit * 2;
        reset();
        double ms = time_kernel(k_fpmul_fips, grid, block, 3, dio, fit);
        printf("{\"bench\": \"fp_mul_fips\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gmul_s\": %.2f}\n", wps, ms, nmul / ms / 1e6);
        reset();
        ms = time_kernel(k_fqmul, grid, block, 3, dio, fit);
        printf("{\"bench\": \"fp_mul_signed30\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gmul_s\": %.2f}\n", wps, ms, nmul / ms / 1e6);
        reset();
        ms = time_kernel(k_fpsqr_fips, grid, block, 3, dio, 2 * fit);
        printf("{\"bench\": \"fp_sqr_fips\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gsqr_s\": %.2f}\n", wps, ms, nmul / ms / 1e6);
        reset();
        ms = time_kernel(k_fqsqr, grid, block, 3, dio, 2 * fit);
        printf("{\"bench\": \"fp_sqr_signed30\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gsqr_s\": %.2f}\n", wps, ms, nmul / ms / 1e6);
        ms = time_kernel(k_dpf_mix, grid, block, 3, out, fit);
        printf("{\"bench\": \"fp_mul_fp64_instruction_mix\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gmul_s_equivalent\": %.2f, \"note\": \"2 v_fma_f64 + v_add_f64 + 2 u64 adds (+1 chaining fma) x 128 limb products; not a multiplier\"}\n",
               wps, ms, (double)grid * block * fit / ms / 1e6);
        const int mit = 64;
        double nadd = (double)grid * block * mit;
        reset();
        ms = time_kernel(k_madd, grid, block, 3, dio, mit);
        printf("{\"bench\": \"xyzz_madd_fips\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gadd_s\": %.3f}\n", wps, ms, nadd / ms / 1e6);
        reset();
        ms = time_kernel(k_madd30, grid, block, 3, dio, mit);
        printf("{\"bench\": \"xyzz_madd_signed30\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gadd_s\": %.3f}\n", wps, ms, nadd / ms / 1e6);
        fflush(stdout);
    }
    return 0;
}


static int run_quad() {
    const u32 GX[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                        0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
    const u32 GY[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                        0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
    u32 hin[256];
    memset(hin, 0, sizeof hin);
    memcpy(hin, GX, 48);
    memcpy(hin + 12, GY, 48);
    u32* dio;
    int* cmp;
    CHECK(hipMalloc(&dio, sizeof hin));
    CHECK(hipMalloc(&cmp, 1024));
    int hc[256];
    CHECK(hipMemset(cmp, 0, 1024));
    hipLaunchKernelGGL(k_dpp_probe, dim3(1), dim3(64), 0, 0, cmp);
    CHECK(hipMemcpy(hc, cmp, 256, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"dpp_quad_broadcast_lane2\", \"first8\": [%d, %d, %d, %d, %d, %d, %d, %d]}\n", hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7]);
    for (int iters = 1; iters <= 4; iters++) {
        CHECK(hipMemcpy(dio, hin, sizeof hin, hipMemcpyHostToDevice));
        CHECK(hipMemset(cmp, 0, 1024));
        hipLaunchKernelGGL(k_addquad, dim3(2), dim3(256), 0, 0, dio, iters, cmp);
        CHECK(hipMemcpy(hc, cmp, 96 * 4, hipMemcpyDeviceToHost));
        printf("{\"probe\": \"add_quad_vs_add\", \"iters\": %d, \"lanes_point_differs\": %d, \"lanes_digits_differ\": %d}\n", iters, hc[0], hc[1]);
        if (hc[0])  // per-lane detail of the first quads: which coordinates (bit 0 X, 1 Y, 2 ZZ, 3 ZZZ) agree digit for digit
            for (int k = 0; k < 16; k++) printf("  lane %d mask %x quadY0 %d quadY12 %d refY0 %d refY12 %d\n", k, hc[8 + k], hc[32 + k], hc[48 + k], hc[64 + k], hc[80 + k]);
    }
    {
        unsigned long long* st;
        static unsigned long long hs[8];
        CHECK(hipMalloc(&st, sizeof hs));
        for (int mode = 0; mode < 2; mode++)
            for (int active : {16, 8, 4, 2, 1}) {
                CHECK(hipMemcpy(dio, hin, sizeof hin, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k_partial_probe, dim3(1), dim3(64), 0, 0, dio, st, active, mode);
                CHECK(hipMemcpy(hs, st, sizeof hs, hipMemcpyDeviceToHost));
                printf("{\"probe\": \"partially_active_wave\", \"mode\": %d, \"active_quads\": %d, \"add_us\": [%.1f, %.1f, %.1f, %.1f]}\n", mode, active,
                       (hs[1] - hs[0]) / 100.0, (hs[3] - hs[2]) / 100.0, (hs[5] - hs[4]) / 100.0, (hs[7] - hs[6]) / 100.0);
            }
    }
    {
        unsigned long long* st;
        static unsigned long long hs[8 * 4];
        CHECK(hipMalloc(&st, sizeof hs));
        for (int waves : {1, 2, 4})
            for (int skew : {0, 1, 3, 6}) {
                CHECK(hipMemcpy(dio, hin, sizeof hin, hipMemcpyHostToDevice));
                CHECK(hipMemset(st, 0, sizeof hs));
                hipLaunchKernelGGL(k_phase_probe, dim3(1), dim3(256), 0, 0, dio, st, skew, waves);
                CHECK(hipMemcpy(hs, st, sizeof hs, hipMemcpyDeviceToHost));
                printf("{\"probe\": \"out_of_phase_waves\", \"active_waves\": %d, \"skew_sleeps\": %d, \"add_us_per_wave\": [", waves, skew);
                for (int w = 0; w < waves; w++) {
                    printf("%s[", w ? ", " : "");
                    for (int it = 0; it < 4; it++) printf("%s%.1f", it ? ", " : "", (hs[w * 8 + it * 2 + 1] - hs[w * 8 + it * 2]) / 100.0);
                    printf("] @%.1f", (hs[w * 8] - hs[0]) / 100.0);
                }
                printf("]}\n");
            }
    }
    {
        unsigned long long* st;
        static unsigned long long hs[8 * 256];
        CHECK(hipMalloc(&st, sizeof hs));
        for (int wgs : {1, 56})
            for (int rep : {0, 2, 5, 3, 4}) {  // rep = mode
                CHECK(hipMemcpy(dio, hin, sizeof hin, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(k_tree_probe, dim3(wgs), dim3(256), 0, 0, dio, st, rep);
                CHECK(hipMemcpy(hs, st, sizeof hs, hipMemcpyDeviceToHost));
                printf("{\"probe\": \"tree_levels_us\", \"workgroups\": %d, \"mode\": %d, \"min_med_max_per_level\": [", wgs, rep);
                for (int l = 0; l < 6; l++) {
                    std::vector<double> v;
                    for (int w = 0; w < wgs; w++) v.push_back((hs[w * 8 + l + 1] - hs[w * 8 + l]) / 100.0);
                    std::sort(v.begin(), v.end());
                    printf("%s[%.1f, %.1f, %.1f]", l ? ", " : "", v[0], v[v.size() / 2], v.back());
                }
                printf("]}\n");
            }
    }
    for (int it = 0; it < 2; it++) {
        CHECK(hipMemcpy(dio, hin, sizeof hin, hipMemcpyHostToDevice));
        double ms = time_kernel(k_addquad, 1, 64, 5, dio, 64, cmp);
        printf("{\"bench\": \"lone_wave_add_plus_add_quad\", \"us_per_pair\": %.3f}\n", ms * 1e3 / 64.0);
    }
    return 0;
}
int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "quad")) return run_quad();
    if (argc > 1 && !strcmp(argv[1], "field30")) return run_field30();
    if (argc > 1 && !strcmp(argv[1], "carrydbg")) return run_carry_debug();
    if (argc > 1 && !strcmp(argv[1], "carry")) return run_carry();
    if (argc > 1 && !strcmp(argv[1], "mix")) return run_mix_all();
    if (argc > 1 && !strcmp(argv[1], "gather")) return run_gather_calibration();
    if (argc > 1 && !strcmp(argv[1], "footprint")) return run_gather_footprints();
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    u64* out;
    CHECK(hipMalloc(&out, sizeof(u64) * 256 * 8 * 1024));
    const int block = 256;
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD
        int grid = cus * wps;
        double n_inst = (double)grid * block * iters * 8 * CHAINS;
        double ms;
        ms = time_kernel(k_mad, grid, block, 5, out, iters);
        printf("{\"bench\": \"v_mad_u64_u32\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gop_s\": %.1f}\n", wps, ms, n_inst / ms / 1e6);
        ms = time_kernel(k_mullo, grid, block, 5, out, iters);
        printf("{\"bench\": \"v_mul_lo_u32\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gop_s\": %.1f}\n", wps, ms, n_inst / ms / 1e6);
        ms = time_kernel(k_mulhi, grid, block, 5, out, iters);
        printf("{\"bench\": \"v_mul_hi_u32(+add)\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gop_s\": %.1f}\n", wps, ms, n_inst / ms / 1e6);
        ms = time_kernel(k_mad24, grid, block, 5, out, iters);
        printf("{\"bench\": \"v_mul_u32_u24(+shift+add)\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gop_s\": %.1f}\n", wps, ms, n_inst / ms / 1e6);
        ms = time_kernel(k_dfma, grid, block, 5, out, iters);
        printf("{\"bench\": \"v_fma_f64\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gop_s\": %.1f}\n", wps, ms, n_inst / ms / 1e6);
        double n_add = (double)grid * block * iters * 8 * 12;
        ms = time_kernel(k_addc, grid, block, 5, out, iters);
        printf("{\"bench\": \"v_addc_co_u32 chain\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gop_s\": %.1f}\n", wps, ms, n_add / ms / 1e6);
    }

    // field-level
    u32 h[128];
    memset(h, 0, sizeof h);
    // x = Gx (Montgomery), y = Gy (Montgomery)
    const u32 GX[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                        0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
    const u32 GY[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                        0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
    u32* dio;
    CHECK(hipMalloc(&dio, sizeof h));
    for (int wps = 1; wps <= 4; wps++) {
        int grid = cus * wps;
        const int fit = 200;
        memcpy(h, GX, 48);
        memcpy(h + 12, GY, 48);
        CHECK(hipMemcpy(dio, h, sizeof h, hipMemcpyHostToDevice));
        double ms = time_kernel(k_fpmul, grid, block, 3, dio, fit);
        double nmul = (double)grid * block * fit * 2;
        printf("{\"bench\": \"fp_mul\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gmul_s\": %.2f}\n", wps, ms, nmul / ms / 1e6);
        ms = time_kernel(k_fpmul_fips, grid, block, 3, dio, fit);
        printf("{\"bench\": \"fp_mul_fips\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gmul_s\": %.2f}\n", wps, ms, nmul / ms / 1e6);
        CHECK(hipMemcpy(dio, h, sizeof h, hipMemcpyHostToDevice));
        ms = time_kernel(k_frmul, grid, block, 3, dio, fit);
        printf("{\"bench\": \"fr_mul\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gmul_s\": %.2f}\n", wps, ms, nmul / ms / 1e6);
        CHECK(hipMemcpy(dio, h, sizeof h, hipMemcpyHostToDevice));
        const int mit = 64;
        ms = time_kernel(k_madd, grid, block, 3, dio, mit);
        double nadd = (double)grid * block * mit;
        printf("{\"bench\": \"xyzz_madd\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gadd_s\": %.3f}\n", wps, ms, nadd / ms / 1e6);
        ms = time_kernel(k_add, grid, block, 3, dio, mit / 2);
        printf("{\"bench\": \"xyzz_add\", \"waves_per_simd\": %d, \"ms\": %.4f, \"Gadd_s\": %.3f}\n", wps, ms, nadd / ms / 1e6);
    }
    // correctness probes (checked offline against Python big ints): 3 iterations of the fpmul loop
    memcpy(h, GX, 48);
    memcpy(h + 12, GY, 48);
    CHECK(hipMemcpy(dio, h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fpmul, dim3(1), dim3(64), 0, 0, dio, 3);
    CHECK(hipMemcpy(h, dio, sizeof h, hipMemcpyDeviceToHost));
    {
        u32 h2[128];
        memset(h2, 0, sizeof h2);
        memcpy(h2, GX, 48);
        memcpy(h2 + 12, GY, 48);
        CHECK(hipMemcpy(dio, h2, sizeof h2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_fpmul_fips, dim3(1), dim3(64), 0, 0, dio, 3);
        CHECK(hipMemcpy(h2, dio, sizeof h2, hipMemcpyDeviceToHost));
        printf("{\"probe\": \"fpmul3_fips\", ");
        hexout("x", h2 + 24, 12);
        printf(", ");
        hexout("y", h2 + 36, 12);
        printf("}\n");
    }
    printf("{\"probe\": \"fpmul3\", ");
    hexout("x", h + 24, 12);
    printf(", ");
    hexout("y", h + 36, 12);
    printf("}\n");
    memset(h, 0, sizeof h);
    memcpy(h, GX, 32);
    memcpy(h + 8, GY, 32);
    h[7] &= 0x3fffffffu;
    h[15] &= 0x3fffffffu;
    u32 hin[16];
    memcpy(hin, h, 64);
    CHECK(hipMemcpy(dio, h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_frmul, dim3(1), dim3(64), 0, 0, dio, 3);
    CHECK(hipMemcpy(h, dio, sizeof h, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"frmul3\", ");
    hexout("x0", hin, 8);
    printf(", ");
    hexout("y0", hin + 8, 8);
    printf(", ");
    hexout("x", h + 16, 8);
    printf(", ");
    hexout("y", h + 24, 8);
    printf("}\n");
    memset(h, 0, sizeof h);
    memcpy(h, GX, 48);
    memcpy(h + 12, GY, 48);
    CHECK(hipMemcpy(dio, h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_madd, dim3(1), dim3(64), 0, 0, dio, 5);
    CHECK(hipMemcpy(h, dio, sizeof h, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"madd5\", ");
    hexout("X", h + 24, 12);
    printf(", ");
    hexout("Y", h + 36, 12);
    printf(", ");
    hexout("ZZ", h + 48, 12);
    printf(", ");
    hexout("ZZZ", h + 60, 12);
    printf("}\n");
    return 0;
}
