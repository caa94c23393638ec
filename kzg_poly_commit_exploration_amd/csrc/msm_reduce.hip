// msm_reduce.hip -- bucket reduction: sum_b (b+1) * B_b from the 2^(c-1) bucket sums.
//
// The weights are linear in the bucket index, so the weighted sum separates over any split of the
// index into digits: with b = hi * C + lo,
//        sum_b b B_b = C * sum_hi hi * Row_hi + sum_lo lo * Col_lo,
//        Row_hi = sum_lo B[hi*C + lo],   Col_lo = sum_hi B[hi*C + lo],   sum_b B_b = sum_hi Row_hi.
// Rows and columns are PLAIN sums -> logarithmic-depth trees instead of running-sum chains.  The same
// split is applied once more to the Row and the Col vector (<= 1024 entries -> four vectors of <= 32
// entries), whose short weighted sums the host finishes (api.hip).  Critical path on the device:
// log2(C) + log2(sqrt) dependent 384-bit additions (12 at 65536 buckets) instead of ~70 with running sums;
// the work stays 2 additions per bucket.  Everything here is bound by the latency of the field
// multiplier (~27 us per addition for a lone wave), not by HBM or VALU throughput.
//
// Field products stay inlined: measured on MI355X the call-based multiplier costs the general
// addition 40 % (2.8 vs 1.6 G additions/s, gpurun_out microbench), the register traffic around the
// calls outweighing the instruction-cache savings.
#define KZG_FAST_DBL_IN_ADD 1
#include "engine.h"
#include "g1.hip.h"

namespace kzg {

KZG_DEV XYZZ load_xyzz(const uint4* __restrict__ in) {
    XYZZ a;
    Fp* f[4] = {&a.X, &a.Y, &a.ZZ, &a.ZZZ};
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int t = 0; t < 3; t++) {
            uint4 v = in[q * 3 + t];
            f[q]->l[4 * t] = v.x;
            f[q]->l[4 * t + 1] = v.y;
            f[q]->l[4 * t + 2] = v.z;
            f[q]->l[4 * t + 3] = v.w;
        }
    }
    return a;
}
KZG_DEV void store_xyzz(uint4* __restrict__ out, const XYZZ& a) {
    const Fp* f[4] = {&a.X, &a.Y, &a.ZZ, &a.ZZZ};
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int t = 0; t < 3; t++) {
            uint4 v;
            v.x = f[q]->l[4 * t];
            v.y = f[q]->l[4 * t + 1];
            v.z = f[q]->l[4 * t + 2];
            v.w = f[q]->l[4 * t + 3];
            out[q * 3 + t] = v;
        }
    }
}

constexpr int kTreeBlock = 256;

// out[g] = sum_{q < len} in[g * gstride + q * estride]   for g < groups  (strides in XYZZ records).
// A workgroup of 256 lanes serves 256 / lanes_per_group groups; each lane first adds its share of the
// group serially (only when len > 256), then the lanes of a group fold in a tree through LDS.
__global__ void __launch_bounds__(kTreeBlock) k_tree_sum(const uint4* __restrict__ in, uint32_t groups, uint32_t len,
                                                         uint32_t lanes_per_group /* pow2, <= 256 */, uint64_t gstride,
                                                         uint64_t estride, uint4* __restrict__ out) {
    __shared__ u32 lds[48 * kTreeBlock];
    const int t = threadIdx.x;
    const uint32_t gpb = kTreeBlock / lanes_per_group;
    const uint32_t g = blockIdx.x * gpb + t / lanes_per_group;
    const uint32_t l = t & (lanes_per_group - 1);
    XYZZ acc = XYZZ::inf();
    if (g < groups) {
        for (uint32_t q = l; q < len; q += lanes_per_group) {
            XYZZ b = load_xyzz(in + (size_t)(g * gstride + q * estride) * 12);
            xyzz_add(acc, b);
        }
    }
    for (uint32_t off = lanes_per_group >> 1; off >= 1; off >>= 1) {
        __syncthreads();
        if (l >= off && l < 2 * off) {
            const Fp* f[4] = {&acc.X, &acc.Y, &acc.ZZ, &acc.ZZZ};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < 12; i++) lds[(q * 12 + i) * kTreeBlock + (t - off)] = f[q]->l[i];
        }
        __syncthreads();
        if (l < off) {
            XYZZ o;
            Fp* f[4] = {&o.X, &o.Y, &o.ZZ, &o.ZZZ};
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < 12; i++) f[q]->l[i] = lds[(q * 12 + i) * kTreeBlock + t];
            xyzz_add(acc, o);
        }
    }
    if (l == 0 && g < groups) store_xyzz(out + (size_t)g * 12, acc);
}

void launch_tree_sum(hipStream_t s, const void* d_in, uint32_t groups, uint32_t len, uint64_t gstride, uint64_t estride,
                     void* d_out) {
    if (!groups) return;
    uint32_t lpg = 1;
    while (lpg < len && lpg < (uint32_t)kTreeBlock) lpg <<= 1;
    uint32_t gpb = kTreeBlock / lpg;
    hipLaunchKernelGGL(k_tree_sum, dim3((groups + gpb - 1) / gpb), dim3(kTreeBlock), 0, s,
                       reinterpret_cast<const uint4*>(d_in), groups, len, lpg, gstride, estride,
                       reinterpret_cast<uint4*>(d_out));
}

}  // namespace kzg
