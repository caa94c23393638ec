// msm_accum.hip -- bucket accumulation, the dominant kernel of the commitment.
//
// Replaces the reference's hot loop  commitment.add(srs[i].g1.mult(c_i))  (src/polynomial.rs:208-212):
// after msm_sort every bucket owns a contiguous list of signed references into the window table
// T[j*n+i] = 2^(c*j) SRS[i]; one lane adds its bucket's points into an XYZZ accumulator held in
// VGPRs (8M + 2S per point, 384-bit Montgomery arithmetic on v_mad_u64_u32).
//
// Roofline: VALU (integer multiply) bound.  Per point ~2900 v_mad_u64_u32 + ~6000 other VALU ops
// against 96 B fetched from the table + 4 B of reference.  Algorithmic HBM bytes per commitment are
// those of SURVEY.md section 8(d): 128 B x n + 144 B.
//
// Load balance: bucket populations are Poisson-like, so a workgroup first orders its 256 buckets by
// population in LDS (counting sort on LDS atomics); each wavefront then owns 64 buckets of nearly
// equal length and its lanes stay converged.
#include "engine.h"
#include "g1.hip.h"

namespace kzg {

constexpr int kAccumBlock = 256;
constexpr int kCountBins = 1024;

KZG_DEV Affine load_affine(const uint4* __restrict__ table, uint32_t idx) {
    const uint4* p = table + (size_t)idx * 6;
    uint4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5];
    Affine r;
    r.x.l[0] = a0.x; r.x.l[1] = a0.y; r.x.l[2] = a0.z; r.x.l[3] = a0.w;
    r.x.l[4] = a1.x; r.x.l[5] = a1.y; r.x.l[6] = a1.z; r.x.l[7] = a1.w;
    r.x.l[8] = a2.x; r.x.l[9] = a2.y; r.x.l[10] = a2.z; r.x.l[11] = a2.w;
    r.y.l[0] = a3.x; r.y.l[1] = a3.y; r.y.l[2] = a3.z; r.y.l[3] = a3.w;
    r.y.l[4] = a4.x; r.y.l[5] = a4.y; r.y.l[6] = a4.z; r.y.l[7] = a4.w;
    r.y.l[8] = a5.x; r.y.l[9] = a5.y; r.y.l[10] = a5.z; r.y.l[11] = a5.w;
    return r;
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

__global__ void __launch_bounds__(kAccumBlock) k_bucket_accumulate(const uint4* __restrict__ table,
                                                                  const uint32_t* __restrict__ sorted,
                                                                  const uint32_t* __restrict__ offs, uint32_t nb,
                                                                  uint4* __restrict__ buckets) {
    __shared__ u32 s_hist[kCountBins];
    __shared__ u32 s_order[kAccumBlock];
    const int t = threadIdx.x;
    const uint32_t b0 = blockIdx.x * kAccumBlock;

    // ---- order this workgroup's buckets by decreasing population ----
    for (int q = t; q < kCountBins; q += kAccumBlock) s_hist[q] = 0;
    __syncthreads();
    uint32_t my_b = b0 + t;
    u32 beg = 0, cnt = 0;
    if (my_b < nb) {
        beg = offs[my_b];
        cnt = offs[my_b + 1] - beg;
    }
    u32 key = cnt < (u32)kCountBins - 1 ? cnt : (u32)kCountBins - 1;
    key = (kCountBins - 1) - key;  // descending
    u32 myrank = atomicAdd(&s_hist[key], 1u);
    __syncthreads();
    // exclusive scan of 1024 bins by 256 threads (4 bins each + block scan)
    {
        u32 v0 = s_hist[4 * t], v1 = s_hist[4 * t + 1], v2 = s_hist[4 * t + 2], v3 = s_hist[4 * t + 3];
        u32 sum = v0 + v1 + v2 + v3;
        __shared__ u32 s_scan[kAccumBlock];
        s_scan[t] = sum;
        __syncthreads();
        for (int off = 1; off < kAccumBlock; off <<= 1) {
            u32 add = t >= off ? s_scan[t - off] : 0u;
            __syncthreads();
            s_scan[t] += add;
            __syncthreads();
        }
        u32 ex = s_scan[t] - sum;
        s_hist[4 * t] = ex;
        s_hist[4 * t + 1] = ex + v0;
        s_hist[4 * t + 2] = ex + v0 + v1;
        s_hist[4 * t + 3] = ex + v0 + v1 + v2;
        __syncthreads();
    }
    s_order[s_hist[key] + myrank] = (u32)t;
    __syncthreads();
    const int src = (int)s_order[t];  // lane t takes over the bucket first seen by lane `src`
    // fetch (beg, cnt) of the adopted bucket
    __shared__ u32 s_beg[kAccumBlock], s_cnt[kAccumBlock];
    s_beg[t] = beg;
    s_cnt[t] = cnt;
    __syncthreads();
    beg = s_beg[src];
    cnt = s_cnt[src];
    const uint32_t b = b0 + (uint32_t)src;
    if (b >= nb) return;

    // ---- accumulate ----
    XYZZ acc = XYZZ::inf();
    for (u32 e = 0; e < cnt; e++) {
        u32 ref = sorted[beg + e];
        Affine p = load_affine(table, ref & 0x7fffffffu);
        xyzz_madd(acc, p, (ref >> 31) != 0);
    }
    store_xyzz(buckets + (size_t)b * 12, acc);
}

void launch_bucket_accumulate(hipStream_t s, const void* d_table, const uint32_t* d_sorted, const uint32_t* d_offs,
                              const uint32_t*, uint32_t nb, void* d_buckets) {
    hipLaunchKernelGGL(k_bucket_accumulate, dim3((nb + kAccumBlock - 1) / kAccumBlock), dim3(kAccumBlock), 0, s,
                       reinterpret_cast<const uint4*>(d_table), d_sorted, d_offs, nb,
                       reinterpret_cast<uint4*>(d_buckets));
}

}  // namespace kzg
