// msm_finalize.hip -- completes the buckets whose references were split over several accumulation
// segments (msm_accum.hip): adds the head / tail partials.  Kept in its own translation unit so that
// its register allocation cannot perturb the dominant kernel's (co-compiled kernels share the
// allocator's context; k_bucket_accumulate must stay at 4 waves per SIMD).
// Bound by the latency of the field multiplier (a handful of dependent additions per bucket).
#include "engine.h"
#include "g1.hip.h"

namespace kzg {

constexpr uint32_t kSerialSpan = 48;  // buckets spanning more segments than this go to the tree kernel
constexpr int kHeavyBlock = 256;
constexpr int kHeavyGrid = 512;

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

// Buckets that span several segments: add up their partials (first segment's tail or whole, whole
// middle segments, last segment's head).  One lane per bucket; very long spans (skewed scalars) are
// queued for k_bucket_heavy.
__global__ void __launch_bounds__(64) k_bucket_finalize(const uint32_t* __restrict__ offs, uint32_t nb, uint32_t lanes,
                                                        const uint4* __restrict__ part_a,
                                                        const uint4* __restrict__ part_b,
                                                        uint4* __restrict__ buckets,
                                                        uint32_t* __restrict__ heavy_list,
                                                        uint32_t* __restrict__ heavy_count) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint32_t L = accumulate_seg_len(offs[nb], lanes);
    uint32_t s = offs[b], e = offs[b + 1];
    if (s == e) return;  // empty bucket: stays at infinity (buffer pre-zeroed)
    uint32_t l_lo = s / L, l_hi = (e - 1) / L;
    if (l_lo == l_hi) return;  // inside one segment: written complete by k_bucket_accumulate
    if (l_hi - l_lo + 1 > kSerialSpan) {
        uint32_t slot = atomicAdd(heavy_count, 1u);
        heavy_list[slot] = b;
        return;
    }
    const uint4* first = (s == l_lo * L) ? part_a + (size_t)l_lo * 12 : part_b + (size_t)l_lo * 12;
    XYZZ acc = load_xyzz(first);
    for (uint32_t l = l_lo + 1; l <= l_hi; l++) {
        XYZZ p = load_xyzz(part_a + (size_t)l * 12);
        xyzz_add(acc, p);
    }
    store_xyzz(buckets + (size_t)b * 12, acc);
}

// One workgroup per queued bucket: strided partial sums, then a tree in LDS.
__global__ void __launch_bounds__(kHeavyBlock) k_bucket_heavy(const uint32_t* __restrict__ offs, uint32_t nb, uint32_t lanes,
                                                              const uint4* __restrict__ part_a,
                                                              const uint4* __restrict__ part_b,
                                                              uint4* __restrict__ buckets,
                                                              const uint32_t* __restrict__ heavy_list,
                                                              const uint32_t* __restrict__ heavy_count) {
    __shared__ u32 lds[48 * kHeavyBlock];
    const int t = threadIdx.x;
    const uint32_t count = *heavy_count;
    const uint32_t L = accumulate_seg_len(offs[nb], lanes);
    for (uint32_t h = blockIdx.x; h < count; h += gridDim.x) {
        uint32_t b = heavy_list[h];
        uint32_t s = offs[b], e = offs[b + 1];
        uint32_t l_lo = s / L, l_hi = (e - 1) / L;
        XYZZ acc = XYZZ::inf();
        for (uint32_t l = l_lo + t; l <= l_hi; l += kHeavyBlock) {
            const uint4* src = (l == l_lo && s != l_lo * L) ? part_b + (size_t)l * 12 : part_a + (size_t)l * 12;
            XYZZ p = load_xyzz(src);
            xyzz_add(acc, p);
        }
        for (int off = kHeavyBlock / 2; off >= 1; off >>= 1) {
            __syncthreads();
            if (t >= off && t < 2 * off) {
                const Fp* f[4] = {&acc.X, &acc.Y, &acc.ZZ, &acc.ZZZ};
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int i = 0; i < 12; i++) lds[(q * 12 + i) * kHeavyBlock + (t - off)] = f[q]->l[i];
            }
            __syncthreads();
            if (t < off) {
                XYZZ o;
                Fp* f[4] = {&o.X, &o.Y, &o.ZZ, &o.ZZZ};
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int i = 0; i < 12; i++) f[q]->l[i] = lds[(q * 12 + i) * kHeavyBlock + t];
                xyzz_add(acc, o);
            }
        }
        if (t == 0) store_xyzz(buckets + (size_t)b * 12, acc);
        __syncthreads();
    }
}

void launch_bucket_finalize(hipStream_t s, const uint32_t* d_offs, uint32_t nb, uint32_t lanes, const void* d_part_a,
                            const void* d_part_b, void* d_buckets, uint32_t* d_heavy_list, uint32_t* d_heavy_count) {
    hipLaunchKernelGGL(k_bucket_finalize, dim3((nb + 63) / 64), dim3(64), 0, s, d_offs, nb, lanes,
                       reinterpret_cast<const uint4*>(d_part_a), reinterpret_cast<const uint4*>(d_part_b),
                       reinterpret_cast<uint4*>(d_buckets), d_heavy_list, d_heavy_count);
    hipLaunchKernelGGL(k_bucket_heavy, dim3(kHeavyGrid), dim3(kHeavyBlock), 0, s, d_offs, nb, lanes,
                       reinterpret_cast<const uint4*>(d_part_a), reinterpret_cast<const uint4*>(d_part_b),
                       reinterpret_cast<uint4*>(d_buckets), d_heavy_list, d_heavy_count);
}

}  // namespace kzg
