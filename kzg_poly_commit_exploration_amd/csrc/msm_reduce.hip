// msm_reduce.hip -- bucket reduction: sum_b (b+1) * B_b from the 2^(c-1) bucket sums.
//
// Level structure (host-driven, see api.hip): a level takes items I[0..n) with weights 1..n, cuts
// them into chunks of m and lets one lane per chunk walk its chunk from the top with the classic
// running sum (run += I; acc += run):
//     acc[k] = sum_t (t+1) I[k m + t]        run[k] = sum_t I[k m + t]
//     sum_i (i+1) I[i] = sum_k acc[k] + m * sum_k k run[k]
// so the next level's items are run[1..] with weights 1.., and the acc[] arrays are plain-summed.
// Everything here is a chain of dependent 384-bit additions at a handful of waves: it is bound by
// the latency of the field multiplier, not by HBM or VALU throughput, and kept short (m small).
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

__global__ void __launch_bounds__(64) k_wsum_level(const uint4* __restrict__ in, uint32_t n_items, uint32_t m,
                                                   uint4* __restrict__ acc_out, uint4* __restrict__ run_out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t chunks = (n_items + m - 1) / m;
    if (k >= chunks) return;
    XYZZ run = XYZZ::inf(), acc = XYZZ::inf();
    for (uint32_t t = m; t-- > 0;) {
        uint32_t idx = k * m + t;
        if (idx < n_items) {
            XYZZ b = load_xyzz(in + (size_t)idx * 12);
            xyzz_add(run, b);
        }
        xyzz_add(acc, run);
    }
    store_xyzz(acc_out + (size_t)k * 12, acc);
    store_xyzz(run_out + (size_t)k * 12, run);
}

__global__ void __launch_bounds__(64) k_sum_level(const uint4* __restrict__ in, uint32_t n_items, uint32_t m,
                                                  uint4* __restrict__ out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t chunks = (n_items + m - 1) / m;
    if (k >= chunks) return;
    XYZZ acc = XYZZ::inf();
    for (uint32_t t = 0; t < m; t++) {
        uint32_t idx = k * m + t;
        if (idx < n_items) {
            XYZZ b = load_xyzz(in + (size_t)idx * 12);
            xyzz_add(acc, b);
        }
    }
    store_xyzz(out + (size_t)k * 12, acc);
}

void launch_wsum_level(hipStream_t s, const void* d_in, uint32_t n_items, uint32_t m, void* d_acc, void* d_run) {
    uint32_t chunks = (n_items + m - 1) / m;
    if (chunks == 0) return;
    hipLaunchKernelGGL(k_wsum_level, dim3((chunks + 63) / 64), dim3(64), 0, s, reinterpret_cast<const uint4*>(d_in),
                       n_items, m, reinterpret_cast<uint4*>(d_acc), reinterpret_cast<uint4*>(d_run));
}
void launch_sum_level(hipStream_t s, const void* d_in, uint32_t n_items, uint32_t m, void* d_out) {
    uint32_t chunks = (n_items + m - 1) / m;
    if (chunks == 0) return;
    hipLaunchKernelGGL(k_sum_level, dim3((chunks + 63) / 64), dim3(64), 0, s, reinterpret_cast<const uint4*>(d_in),
                       n_items, m, reinterpret_cast<uint4*>(d_out));
}

}  // namespace kzg
