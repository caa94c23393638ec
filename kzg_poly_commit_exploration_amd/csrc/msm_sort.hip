// msm_sort.hip -- scalar recoding and the counting sort that turns (scalar, window) pairs into
// bucket-major lists of table references.  Replaces the per-term Scalar::to_le_bytes +
// blst_p1_mult window walk of the reference (src/scalar.rs:83-93, src/curves.rs:90-96) for the
// whole polynomial at once.
//
// HBM traffic per commitment of n terms, W windows: scalars read twice (2 x 32 B x n), ranks
// written + read (2 x 4 B x n x W), references written (4 B x n x W), histogram atomics (n x W).
// This is integer / byte work bound by scattered 4-byte accesses, not by arithmetic.
#include "engine.h"
#include "field.hip.h"

namespace kzg {

MsmConfig choose_msm_config(size_t n) {
    // Scalars are first folded to |k| <= (r-1)/2 < 2^254 (sign moved onto the point), so
    // W = ceil(255 / c) signed windows never carry out of the top.  c minimises
    //     n * W  (mixed additions)  +  8 * 2^(c-1)  (bucket finalisation + reduction, weighted for
    // the latency of the reduction levels): 17 bits / 15 windows / 65536 buckets at 2^20 points.
    uint32_t best_c = 8;
    double best = 1e300;
    for (uint32_t c = 8; c <= 20; c++) {
        uint32_t W = (255 + c - 1) / c;
        double cost = (double)n * W + 8.0 * (double)(1u << (c - 1));
        if (cost < best) {
            best = cost;
            best_c = c;
        }
    }
    MsmConfig cfg;
    cfg.c = best_c;
    cfg.W = (255 + best_c - 1) / best_c;
    cfg.nb = 1u << (best_c - 1);
    return cfg;
}

// canonical 256-bit scalar (8 words) from the stored form, folded to the shorter of k and r - k:
// k * P = (r - k) * (-P).  Returns true when the point has to be negated.  Besides halving the
// range this makes the reference's "negative" i128 inputs (r - |a|, src/scalar.rs:27-48) as cheap
// as the positive ones: their upper windows become zero digits, which are skipped.
KZG_DEV bool load_scalar(const uint32_t* d_scalars, uint32_t i, int is_mont, u32 k[8]) {
    const uint4* p = reinterpret_cast<const uint4*>(d_scalars) + 2 * (size_t)i;
    uint4 lo = p[0], hi = p[1];
    Fr a;
    a.l[0] = lo.x; a.l[1] = lo.y; a.l[2] = lo.z; a.l[3] = lo.w;
    a.l[4] = hi.x; a.l[5] = hi.y; a.l[6] = hi.z; a.l[7] = hi.w;
    if (is_mont) {
        a = fe_from_mont(a);
    } else {
        // canonical little-endian bytes are expected below r; reduce defensively (2^256 < 3r)
        cond_sub_mod(a, 0u);
        cond_sub_mod(a, 0u);
    }
    // neg = r - a ; use it when neg < a  (a > (r-1)/2)
    u32 nk[8];
    u32 br = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) nk[t] = subb(FrParams::mod(t), a.l[t], br);
    // compare nk < a : borrow of nk - a
    u32 b2 = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) (void)subb(nk[t], a.l[t], b2);
    bool flip = b2 != 0;
#pragma unroll
    for (int t = 0; t < 8; t++) k[t] = flip ? nk[t] : a.l[t];
    return flip;
}

// Signed window recoding, low window first: digit in [-2^(c-1)+1, 2^(c-1)], carry into the next
// window.  |k| < 2^254 and W * c >= 255, so the top window absorbs the last carry.
// f(j, magnitude (>0), negative)
template <class F>
KZG_DEV void for_each_digit(u32 k[8], uint32_t c, uint32_t W, F&& f) {
    const u32 mask = (1u << c) - 1u;
    const u32 half = 1u << (c - 1);
    u32 carry = 0;
    for (uint32_t j = 0; j < W; j++) {
        u32 v = (k[0] & mask) + carry;
        // k >>= c
#pragma unroll
        for (int t = 0; t < 7; t++) k[t] = (k[t] >> c) | (k[t + 1] << (32 - c));
        k[7] >>= c;
        bool neg = v > half;
        u32 mag = neg ? (mask + 1u - v) : v;
        carry = neg ? 1u : 0u;
        if (mag) f(j, mag, neg);
    }
}

__global__ void __launch_bounds__(256) k_digits_hist(const uint32_t* __restrict__ d_scalars, int is_mont, uint32_t n,
                                                     uint32_t c, uint32_t W, uint32_t* __restrict__ d_hist,
                                                     uint32_t* __restrict__ d_rank) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 k[8];
    (void)load_scalar(d_scalars, i, is_mont, k);
    for_each_digit(k, c, W, [&](uint32_t j, u32 mag, bool) {
        u32 r = atomicAdd(&d_hist[mag - 1], 1u);
        d_rank[(size_t)j * n + i] = r;
    });
}

__global__ void __launch_bounds__(256) k_scatter(const uint32_t* __restrict__ d_scalars, int is_mont, uint32_t n,
                                                 uint32_t table_stride, uint32_t c, uint32_t W,
                                                 const uint32_t* __restrict__ d_offs,
                                                 const uint32_t* __restrict__ d_rank, uint32_t* __restrict__ d_sorted) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 k[8];
    const bool flip = load_scalar(d_scalars, i, is_mont, k);
    for_each_digit(k, c, W, [&](uint32_t j, u32 mag, bool neg) {
        u32 pos = d_offs[mag - 1] + d_rank[(size_t)j * n + i];
        d_sorted[pos] = (j * table_stride + i) | ((neg != flip) ? 0x80000000u : 0u);
    });
}

// ---- exclusive scan of the histogram (<= 2^20 buckets): local / top / add -----------------
constexpr int kScanBlock = 256;
constexpr int kScanPerThread = 4;
constexpr int kScanTile = kScanBlock * kScanPerThread;

__device__ __forceinline__ u32 block_exclusive_scan_256(u32 v, u32* lds, u32& total) {
    // Hillis-Steele over 256 values in LDS
    int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        u32 add = t >= off ? lds[t - off] : 0u;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    total = lds[255];
    return lds[t] - v;
}

__global__ void __launch_bounds__(kScanBlock) k_scan_local(const uint32_t* __restrict__ d_hist, uint32_t nb,
                                                           uint32_t* __restrict__ d_offs,
                                                           uint32_t* __restrict__ d_block_sums) {
    __shared__ u32 lds[kScanBlock];
    uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
    u32 v[kScanPerThread];
    u32 s = 0;
#pragma unroll
    for (int t = 0; t < kScanPerThread; t++) {
        v[t] = (base + t < nb) ? d_hist[base + t] : 0u;
        s += v[t];
    }
    u32 total;
    u32 ex = block_exclusive_scan_256(s, lds, total);
#pragma unroll
    for (int t = 0; t < kScanPerThread; t++) {
        if (base + t < nb) d_offs[base + t] = ex;
        ex += v[t];
    }
    if (threadIdx.x == 0) d_block_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024) k_scan_top(uint32_t* __restrict__ d_block_sums, uint32_t nblocks,
                                                   uint32_t* __restrict__ d_total_out) {
    __shared__ u32 lds[1024];
    int t = threadIdx.x;
    u32 v = (uint32_t)t < nblocks ? d_block_sums[t] : 0u;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        u32 add = t >= off ? lds[t - off] : 0u;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    if ((uint32_t)t < nblocks) d_block_sums[t] = lds[t] - v;
    if (t == 1023) *d_total_out = lds[1023];
}

__global__ void __launch_bounds__(kScanBlock) k_scan_add(uint32_t* __restrict__ d_offs, uint32_t nb,
                                                         const uint32_t* __restrict__ d_block_sums) {
    uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPerThread;
    u32 add = d_block_sums[blockIdx.x];
#pragma unroll
    for (int t = 0; t < kScanPerThread; t++)
        if (base + t < nb) d_offs[base + t] += add;
}

void launch_digits_hist(hipStream_t s, const uint32_t* d_scalars, int is_mont, uint32_t n, MsmConfig cfg,
                        uint32_t* d_hist, uint32_t* d_rank) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_digits_hist, dim3((n + 255) / 256), dim3(256), 0, s, d_scalars, is_mont, n, cfg.c, cfg.W,
                       d_hist, d_rank);
}

void launch_bucket_scan(hipStream_t s, const uint32_t* d_hist, uint32_t nb, uint32_t* d_offs,
                        uint32_t* d_block_sums) {
    uint32_t nblocks = (nb + kScanTile - 1) / kScanTile;  // <= 1024 for nb <= 2^20
    hipLaunchKernelGGL(k_scan_local, dim3(nblocks), dim3(kScanBlock), 0, s, d_hist, nb, d_offs, d_block_sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, s, d_block_sums, nblocks, d_offs + nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nblocks), dim3(kScanBlock), 0, s, d_offs, nb, d_block_sums);
}

void launch_scatter(hipStream_t s, const uint32_t* d_scalars, int is_mont, uint32_t n, uint32_t table_stride,
                    MsmConfig cfg, const uint32_t* d_offs, const uint32_t* d_rank, uint32_t* d_sorted) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_scatter, dim3((n + 255) / 256), dim3(256), 0, s, d_scalars, is_mont, n, table_stride, cfg.c,
                       cfg.W, d_offs, d_rank, d_sorted);
}

}  // namespace kzg
