// msm_sort.hip -- scalar recoding and the counting sort that turns (scalar, window) pairs into
// bucket-major lists of table references.  Replaces the per-term Scalar::to_le_bytes +
// blst_p1_mult window walk of the reference (src/scalar.rs:83-93, src/curves.rs:90-96) for the
// whole polynomial at once.
//
// HBM traffic per commitment of n terms, W windows: scalars read twice (2 x 32 B x n), pairs written
// once and read twice (3 x 4 B x n x W in the packed form -- table index below 2^24, e.g. degree 2^20 -- else
// 3 x 8 B), references written (4 B x n x W).  Integer / byte work bound by LDS atomics and scattered 4..8-byte
// stores, not by arithmetic.
#include <cstdlib>
#include <cstring>

#include "engine.h"
#include "fr30.hip.h"

namespace kzg {

typedef uint32_t u32;
#ifndef KZG_DEV
#define KZG_DEV __device__ __forceinline__
#endif

static uint32_t ilog2(uint32_t v) {
    uint32_t b = 0;
    while ((1u << b) < v) b++;
    return b;
}

MsmConfig choose_msm_config(size_t n, size_t table_budget_bytes) {
    // Scalars are first folded to |k| <= (r-1)/2 < 2^254 (sign moved onto the point), so signed digits never
    // carry out of bit 254.  The digit width c minimises
    //     n * (digits per scalar)  (mixed additions)  +  20 * buckets.
    // A bucket costs the accumulation a run boundary (flush, walk to the next bucket: divergent) and the finalisation
    // and reduction two general additions on latency-bound kernels.  The weight is fitted to measurements on MI355X
    // (round 2, pipelined throughput, widths forced with KZG_MSM_C): 131073 terms x 8 polynomials per step 15 > 16 > 17
    // (2772 / 2545 / 2362 per s), 262145 x 4: 15 = 16 > 17, 524289 x 2: 16 = 17, 2^20: 17 > 16 (408 / 394),
    // 2^22: 19 > 20 (97.1 / 80.4) -- every one of them says 16 <= weight <= 32; round 1's 8 picked 16 / 17 / 17 / 17 / 20.
    //   windows: ceil(255/c) digits, 2^(c-1) buckets -> 17 bits / 15 windows / 65536 buckets at 2^20 points
    //   NAF:     255/(c+1) digits on average, 2^(c-2) buckets -> 19 bits / 13.1 digits / 131072 buckets
    // The NAF recoding is opt-in (KZG_MSM_RECODE=naf, and its 255-level table must fit the budget): measured on
    // MI355X at 2^20 points it saves 12.8 % of the mixed additions but each one costs 12 % more, because random
    // gathers over a 34 GB table miss the per-CU translation cache 74 % of the time instead of 0.5 % over the
    // 2 GB window table (TCP_UTCL1_TRANSLATION_MISS).  KZG_MSM_C=<bits> overrides c.
    const char* force_mode = std::getenv("KZG_MSM_RECODE");
    const char* force_c = std::getenv("KZG_MSM_C");
    const uint32_t fc = force_c ? (uint32_t)std::strtoul(force_c, nullptr, 10) : 0u;
    const bool naf = force_mode && !std::strcmp(force_mode, "naf") && n * 255ull < 0x80000000ull &&
                     (double)n * 255.0 * (double)kAffineBytes <= (double)table_budget_bytes;
    uint32_t best_c = 8;
    const uint32_t best_mode = naf ? kRecodeNaf : kRecodeWindows;
    double best = 1e300;
    for (uint32_t c = 8; c <= (naf ? 21u : 20u); c++) {
        if (fc >= 8 && fc <= (naf ? 21u : 20u) && c != fc) continue;
        // A top window that holds only one or two bits of the (folded, < 2^254) scalars sends a half or a quarter of ALL
        // points into one or two buckets, whose log-depth trees then sit on the critical path of the job (50-80 us at
        // degree 1000 / 2500, 0.1 ms at 2^14 with width 12 against 13).  Widths 9, 11, 12, 14 and 18 are not considered.
        if (!naf && !fc && 254u - c * ((255 + c - 1) / c - 1) < 4u) continue;
        double cost = naf ? (double)n * (255.0 / (c + 1) + 0.5) + 8.0 * (double)(1u << (c - 2))
                          : (double)n * ((255 + c - 1) / c) + 20.0 * (double)(1u << (c - 1));
        if (cost < best) {
            best = cost;
            best_c = c;
        }
    }
    MsmConfig cfg;
    cfg.recode = best_mode;
    cfg.c = best_c;
    if (best_mode == kRecodeNaf) {
        cfg.W = 255;
        cfg.level_bits = 1;
        cfg.nb = 1u << (best_c - 2);
        cfg.max_digits = (255 + best_c - 1) / best_c + 1;  // non-zero digits are at least c bits apart
    } else {
        cfg.W = (255 + best_c - 1) / best_c;
        cfg.level_bits = best_c;
        cfg.nb = 1u << (best_c - 1);
        cfg.max_digits = cfg.W;
    }
    return cfg;
}

// The scalar as sign and magnitude of its shortest representative: |k| <= r / 2 (+ r / 2^31) < 2^254 in 8 words, and
// k * P = |k| * (+-P).  Returns true when the point has to be negated.  Besides halving the range this makes the
// reference's "negative" i128 inputs (r - |a|, src/scalar.rs:27-48) as cheap as the positive ones: their upper windows
// become zero digits, which are skipped.
// One product in the signed-digit field (fr30.hip.h) leaves the Montgomery form, reduces and centres at once: the blst_fr
// image is x * 2^256, times the single digit 2^14 over the multiplier's 2^270 is x; canonical little-endian bytes (expected
// below r, any 256-bit value accepted) times 2^270 mod r over 2^270 is the value mod r.  The product of balanced
// Montgomery digits is the centred residue up to r / 2^31 -- which representative is used changes the digits, not the sum.
// (Two calls: with the multiplier a compile-time constant the first is a reduction and nine shifts, not 162 multiply-adds.)
KZG_DEV bool load_scalar(const uint32_t* d_scalars, uint64_t i, int is_mont, u32 k[8]) {
    const uint4* p = reinterpret_cast<const uint4*>(d_scalars) + 2 * (size_t)i;
    const uint4 lo = p[0], hi = p[1];
    const uint32_t in[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const Fr30 d = fr30_from_limbs_raw(in);
    const Fr30 v = is_mont ? fr30_mul(d, fr30_small(1 << 14)) : fr30_mul(d, fr30_const_one270());
    return fr30_abs_to_limbs(v, k);
}

// Signed window recoding, low window first: digit in [-2^(c-1)+1, 2^(c-1)], carry into the next
// window.  |k| < 2^254 and W * c >= 255, so the top window absorbs the last carry.
// f(table level, bucket, negative)
template <class F>
KZG_DEV void for_each_window_digit(u32 k[8], uint32_t c, uint32_t W, F&& f) {
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
        if (mag) f(j, mag - 1u, neg);
    }
}

// Width-c non-adjacent form, low bit first, without ever shifting the scalar: K' = (k >> pos) + carry is the
// value still to encode.  K' even -> next bit (the carry is unchanged: bit == carry).  K' odd -> the digit is
// v = (c bits of k at pos) + carry (no overflow: an odd K' means bit0 + carry == 1), taken as v - 2^c when
// v > 2^(c-1) (carry 1), and the next c-1 digits are zero.  Words are walked by an unrolled loop so that the
// scalar stays in registers; zero digits are skipped with one ffs per run.  |k| < 2^254 keeps the last digit at
// bit <= 254.  Digits are odd: bucket (|d| - 1) / 2, weight 2 * bucket + 1.
template <class F>
KZG_DEV void for_each_naf_digit(const u32 k[8], uint32_t c, F&& f) {
    const u32 mask = (1u << c) - 1u;
    const u32 half = 1u << (c - 1);
    u32 carry = 0;
    uint32_t p = 0;  // bit offset inside word t (can exceed 32 after a digit)
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const uint64_t win = ((uint64_t)(t < 7 ? k[t + 1] : 0u) << 32) | k[t];
        while (p < 32) {
            u32 x = (u32)(win >> p);
            if (carry) x = ~x;
            // first position at or after p (inside this word) where bit != carry
            uint32_t z = x ? (uint32_t)__builtin_ctz(x) : 32u;
            if (z >= 32u - p) {
                p = 32;
                break;
            }
            p += z;
            u32 v = ((u32)(win >> p) & mask) + carry;
            const bool neg = v > half;
            const u32 mag = neg ? (mask + 1u - v) : v;
            carry = neg ? 1u : 0u;
            f(32u * (uint32_t)t + p, (mag - 1u) >> 1, neg);
            p += c;
        }
        p -= 32;
    }
}

template <class F>
KZG_DEV void for_each_digit(u32 k[8], MsmConfig cfg, F&& f) {
    if (cfg.recode == kRecodeNaf) for_each_naf_digit(k, cfg.c, f);
    else for_each_window_digit(k, cfg.c, cfg.W, f);
}

// ---- two-level counting sort of the (scalar, window) pairs by bucket -------------------------
// Scattered global atomics cap out near 2e10/s on MI355X (they execute at the memory side), which made
// the histogram the second most expensive kernel.  The sort is therefore staged through LDS, in FIVE launches
// (round 2: twelve -- three of them one-workgroup scans and two three-launch table scans, ~6 us apiece whatever they do):
//   count    every workgroup takes a tile of scalars, recodes them and histograms the COARSE bin
//            (bucket >> fine_bits) of each digit in LDS, then takes its range inside every bin with one global atomic
//            per bin (the offset goes out tile-major [tile][bin]; the fill counters end up as the bin totals)
//   spread   every workgroup derives its write cursors: a scan over the <= 2048 bin totals in LDS + its offsets (what
//            two three-launch table scans used to prepare); workgroup 0 also leaves the bin starts and the chunk plan
//            of the fine passes.  Then the same recoding; LDS cursors hand out positions; (fine key, table reference)
//            pairs are written into their coarse bin
//   fine count   the pairs of every coarse bin, in chunks: LDS histogram of the fine key -> H[bin][key][chunk]
//   bin scan     one workgroup per bin: bins of more than kFineLocal chunks (skewed inputs only) get their H rows
//                scanned into global positions here; empty bins get their bucket offsets
//   fine scatter chunk (bin, j): positions from H -- for ordinary bins computed on the spot from the bin's own
//                <= 256 x 16 entries -- then the references are moved to their final, bucket-major position (below);
//                chunk 0 of a bin writes the bin's bucket offsets
// No global atomic per reference (one per tile and bin), no rank array; order inside a bucket is arbitrary (the group law is commutative,
// the result is bit-identical).
constexpr int kSortBlock = 256;
// The two recoding passes run 1024 lanes per workgroup (512 in the staged form of the second): a tile is one workgroup,
// and with 256 lanes a CU held ONE wave per SIMD walking 16 scalars one after the other -- load, from-Montgomery
// product, 15 LDS atomics -- with nothing to hide the loads behind (54 + 85 us at 2^20).
constexpr int kRecodeBlock = 1024;
// Tiles (= workgroups of the two recoding passes) per job, at most: they bound the offset table [tile][bin].  512 puts two
// workgroups on every CU (256: 35 + 87 us for the two passes at 2^20, 512: 35 + 69, 1024: 41 + 78).
#ifndef KZG_SORT_TILES
#define KZG_SORT_TILES 512
#endif
constexpr uint32_t kMaxTiles = KZG_SORT_TILES;
constexpr int kMaxCoarse = 2048;
constexpr int kFineMax = 256;

struct SortGeom {
    uint32_t tile;         // scalars per workgroup in passes 1/2
    uint32_t tiles;        // workgroups
    uint32_t fine_bits;    // low bits of the bucket id resolved in pass 3
    uint32_t coarse_bins;  // nb >> fine_bits
    bool packed;           // pairs are 4 bytes (sign | fine key : 7 | table index : 24) instead of 8
};

// The (fine key, table reference) pairs of the spread pass are written once and read twice: at 8 bytes each they are
// the largest stream of the sort.  When the table index fits 24 bits (W * table stride <= 2^24: a degree-2^20 commitment
// has 15 x (2^20 + 1)) and the buckets split into at most kMaxCoarse bins of 128, a pair is ONE word.
constexpr uint32_t kPackedFineBits = 7, kPackedIndexBits = 24;
static uint32_t sort_fine_bits(MsmConfig cfg, bool packed) {
    const uint32_t bits = ilog2(cfg.nb), most = packed ? kPackedFineBits : 8u;
    return bits < most ? bits : most;
}
static bool sort_packed_buckets(uint32_t nb_total, MsmConfig cfg);

static bool sort_packed_buckets(uint32_t nb_total, MsmConfig cfg) { return (nb_total >> sort_fine_bits(cfg, true)) <= (uint32_t)kMaxCoarse; }

// `total` scalars (batch * n) into `nb_total` buckets (batch * 2^(c-1): polynomial-major bucket ids)
static SortGeom sort_geometry(uint64_t total, uint32_t nb_total, MsmConfig cfg, uint32_t table_stride) {
    SortGeom g;
    static const bool allow_packed = [] { const char* v = std::getenv("KZG_SORT_PACKED"); return !(v && v[0] == '0'); }();
    g.packed = allow_packed && sort_packed_buckets(nb_total, cfg) && (uint64_t)cfg.W * table_stride <= (1ull << kPackedIndexBits);
    uint32_t tile = (uint32_t)((total + kMaxTiles - 1) / kMaxTiles);  // at most kMaxTiles tiles (bounds the count table)
    tile = ((tile + kSortBlock - 1) / kSortBlock) * kSortBlock;
    if (tile < (uint32_t)kSortBlock) tile = kSortBlock;
    g.tile = tile;
    g.tiles = (uint32_t)((total + tile - 1) / tile);
    g.fine_bits = sort_fine_bits(cfg, g.packed);
    g.coarse_bins = nb_total >> g.fine_bits;
    return g;
}

uint32_t sort_max_batch(MsmConfig cfg) {
    uint32_t fine_bits = sort_fine_bits(cfg, false);
    uint32_t b = (uint32_t)kMaxCoarse / (cfg.nb >> fine_bits);
    return b < 1 ? 1 : b;
}

uint32_t sort_count_entries(uint32_t max_batch, MsmConfig cfg) {
    // coarse bins x tiles, tiles bounded by 256 (+1) whatever the length of the commitment
    // (the largest bin count any batch <= max_batch can have: small batches take the packed form, with twice the bins)
    uint32_t bins = 0;
    for (uint32_t b = 1; b <= max_batch; b++) {
        const uint32_t nbt = cfg.nb * b;
        const uint32_t q = nbt >> sort_fine_bits(cfg, sort_packed_buckets(nbt, cfg));
        if (q > bins) bins = q;
    }
    return bins * (kMaxTiles + 1u);
}

// Batch addressing shared by passes 1 and 2: global scalar index g -> polynomial p = g / n, term i = g % n;
// coefficients of polynomial p start at d_scalars + p * stride (in scalars); its buckets at p * nb.
struct BatchGeom {
    uint32_t n;       // terms per polynomial
    uint32_t batch;   // polynomials
    uint64_t stride;  // scalars between consecutive polynomials
    uint32_t nb;      // buckets per polynomial
};

__global__ void __launch_bounds__(kRecodeBlock) k_sort_count(const uint32_t* __restrict__ d_scalars, int is_mont,
                                                           BatchGeom bg, MsmConfig cfg, uint32_t tile,
                                                           uint32_t fine_bits, uint32_t coarse_bins,
                                                           uint32_t* __restrict__ d_cnt, uint32_t* __restrict__ d_binfill /* zero */,
                                                           uint32_t* __restrict__ d_header) {
    __shared__ u32 s_hist[kMaxCoarse];
    if (blockIdx.x == 0 && threadIdx.x < kHeavyHeaderBytes / 4) d_header[threadIdx.x] = 0;  // the job's counters (one stream operation fewer per job)
    for (uint32_t q = threadIdx.x; q < coarse_bins; q += kRecodeBlock) s_hist[q] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * tile;
    const uint64_t total = (uint64_t)bg.n * bg.batch;
    for (uint32_t off = threadIdx.x; off < tile; off += kRecodeBlock) {
        uint64_t g = base + off;
        if (g >= total) break;
        uint32_t p = (uint32_t)(g / bg.n);
        uint32_t i = (uint32_t)(g - (uint64_t)p * bg.n);
        u32 k[8];
        (void)load_scalar(d_scalars, p * bg.stride + i, is_mont, k);
        const u32 pb = p * bg.nb;
        for_each_digit(k, cfg, [&](uint32_t, u32 bkt, bool) { atomicAdd(&s_hist[(pb + bkt) >> fine_bits], 1u); });
    }
    __syncthreads();
    // Every tile takes its range inside every coarse bin with ONE atomic per bin (256 ... 2048 per workgroup, each
    // returning where the tile's pairs of that bin start): the bin fill counters end up holding the bin totals, and no
    // kernel has to add up the count table afterwards.  Which tile comes first inside a bin is whatever order the
    // atomics arrive in -- the order of the references of one bucket is arbitrary anyway.
    for (uint32_t q = threadIdx.x; q < coarse_bins; q += kRecodeBlock) {
        const u32 c = s_hist[q];
        d_cnt[(size_t)blockIdx.x * coarse_bins + q] = c ? atomicAdd(&d_binfill[q], c) : 0u;  // [tile][bin]
    }
}

// exclusive scan of N values, one per thread of an N-lane workgroup, through LDS (Hillis-Steele); their total in `total`
template <int N>
__device__ __forceinline__ u32 block_exclusive_scan_n(u32 v, u32* lds, u32& total);
__device__ __forceinline__ u32 block_exclusive_scan_256(u32 v, u32* lds, u32& total);

// pairs per chunk of the fine passes, and the workspace they share (u32 words)
constexpr uint32_t kFineMaxChunks = 4096;            // H has kFineMaxChunks * 256 <= 2^20 entries
constexpr uint32_t kFineLocal = 16;                  // bins of up to this many chunks are scanned by their scatter workgroups
constexpr uint32_t kWsBinStart = 0;                  // kMaxCoarse + 1 entries: first pair of every coarse bin, then the total
constexpr uint32_t kWsPrefix = kMaxCoarse + 8;       // kMaxCoarse + 1 entries: chunks in front of every bin, then their number
constexpr uint32_t kWsBinFill = 2 * kMaxCoarse + 32; // kMaxCoarse entries, zero between jobs: the bins' fill counters (sort_workspace_zero_words)
constexpr uint32_t kWsTable = 3 * kMaxCoarse + 64;   // H
uint32_t sort_workspace_words() { return kWsTable + kFineMaxChunks * (uint32_t)kFineMax + 64; }
uint32_t sort_workspace_zero_words() { return kWsTable; }  // what the owner clears once, when it allocates the workspace

// cursors of a tile: start of every bin (exclusive scan of the bin totals, <= 2048 values, in LDS) + the offset the tile took
// inside the bin (k_sort_count).  Thread t handles the bins t + BLOCK s.  Workgroup 0 also leaves what the fine passes need:
// the bin starts and their chunk plan.
template <int BLOCK>
KZG_DEV void spread_tile_cursors(uint32_t coarse_bins, const uint32_t* __restrict__ d_cnt, const uint32_t* __restrict__ d_binfill,
                                 uint32_t ch, uint32_t* __restrict__ d_binstart, uint32_t* __restrict__ d_prefix,
                                 uint32_t* __restrict__ d_total, u32* s_cur /* kMaxCoarse */, u32* s_scan /* BLOCK */) {
    const uint32_t t = threadIdx.x;
    constexpr uint32_t kStripes = kMaxCoarse / BLOCK;
    u32 below[kStripes], all[kStripes];
#pragma unroll
    for (uint32_t s = 0; s < kStripes; s++) {
        const uint32_t q = t + s * BLOCK;
        all[s] = q < coarse_bins ? d_binfill[q] : 0u;
        below[s] = q < coarse_bins ? d_cnt[(size_t)blockIdx.x * coarse_bins + q] : 0u;
    }
    u32 carry = 0, chunk_carry = 0;
#pragma unroll
    for (uint32_t s = 0; s < kStripes; s++) {
        if (s * BLOCK >= coarse_bins) break;  // (uniform)
        const uint32_t q = t + s * BLOCK;
        u32 stripe_total;
        const u32 start = carry + block_exclusive_scan_n<BLOCK>(all[s], s_scan, stripe_total);
        __syncthreads();
        if (q < coarse_bins) s_cur[q] = start + below[s];
        if (blockIdx.x == 0) {
            const u32 nch = q < coarse_bins ? (all[s] + ch - 1) / ch : 0u;
            u32 chunk_total;
            const u32 cstart = chunk_carry + block_exclusive_scan_n<BLOCK>(nch, s_scan, chunk_total);
            __syncthreads();
            if (q < coarse_bins) {
                d_binstart[q] = start;
                d_prefix[q] = cstart;
            }
            chunk_carry += chunk_total;
        }
        carry += stripe_total;
    }
    if (blockIdx.x == 0 && t == 0) {
        d_binstart[coarse_bins] = carry;
        d_prefix[coarse_bins] = chunk_carry;
        *d_total = carry;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(kRecodeBlock) k_sort_spread(const uint32_t* __restrict__ d_scalars, int is_mont,
                                                            BatchGeom bg, uint32_t table_stride, MsmConfig cfg,
                                                            uint32_t tile, uint32_t tiles, uint32_t fine_bits,
                                                            uint32_t coarse_bins, const uint32_t* __restrict__ d_cnt /* [tile][bin] */,
                                                            const uint32_t* __restrict__ d_binfill /* bin totals */,
                                                            uint32_t ch, uint32_t* __restrict__ d_binstart,
                                                            uint32_t* __restrict__ d_prefix, uint32_t* __restrict__ d_total,
                                                            uint64_t* __restrict__ d_pairs, int packed) {
    __shared__ u32 s_cur[kMaxCoarse];
    __shared__ u32 s_scan[kRecodeBlock];
    spread_tile_cursors<kRecodeBlock>(coarse_bins, d_cnt, d_binfill, ch, d_binstart, d_prefix, d_total, s_cur, s_scan);
    const uint64_t base = (uint64_t)blockIdx.x * tile;
    const uint64_t total = (uint64_t)bg.n * bg.batch;
    const u32 fine_mask = (1u << fine_bits) - 1u;
    for (uint32_t off = threadIdx.x; off < tile; off += kRecodeBlock) {
        uint64_t g = base + off;
        if (g >= total) break;
        uint32_t p = (uint32_t)(g / bg.n);
        uint32_t i = (uint32_t)(g - (uint64_t)p * bg.n);
        u32 k[8];
        const bool flip = load_scalar(d_scalars, p * bg.stride + i, is_mont, k);
        const u32 pb = p * bg.nb;
        for_each_digit(k, cfg, [&](uint32_t j, u32 bkt, bool neg) {
            u32 b = pb + bkt;
            u32 pos = atomicAdd(&s_cur[b >> fine_bits], 1u);
            u32 ref = (j * table_stride + i) | ((neg != flip) ? 0x80000000u : 0u);
            if (packed) reinterpret_cast<u32*>(d_pairs)[pos] = ref | ((b & fine_mask) << kPackedIndexBits);
            else d_pairs[pos] = ((uint64_t)(b & fine_mask) << 32) | ref;
        });
    }
}

// ---- the spread pass staged through LDS (one-word pairs) ---------------------------------------------------------------
// The direct form above issues one 4-byte store per digit, each lane of a wave into a different bin: 15.7 M store
// transactions per degree-2^20 commitment (84 us whether a pair is 4 or 8 bytes, and however many tiles there are).  Here a
// workgroup takes its tile in rounds of kStageBlock scalars: the round's pairs are counted per bin (LDS atomics), the counts
// scanned, the pairs placed bin-major into an LDS buffer (the atomics' return values are the slots), and written out by
// consecutive lanes -- a bin's pairs of one round are contiguous in the bin (the tile's range inside it is), so a wave's 64
// stores fall into a few runs instead of 64 places.  The digits are cut twice per round (the scalar itself is converted
// once); order inside a bin is arbitrary as before.  69 us at 2^20 with two workgroups per CU (87 with one: a round is a
// chain of a global load, a product and two passes of LDS atomics, and eight waves per CU do not hide it).
constexpr int kStageBlock = 512;
constexpr uint32_t kStageSlots = 7680;  // pairs per round: 512 scalars x 15 digits (fewer scalars per round when a scalar has more)

// exclusive scan of one value per thread: shuffles inside the wave, the wave totals through LDS (two barriers)
template <int BLOCK>
KZG_DEV u32 block_exclusive_scan_fast(u32 v, u32* s_wave /* BLOCK / 64 */, u32& total) {
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    u32 incl = v;
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const u32 up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    u32 base = 0, all = 0;
#pragma unroll
    for (uint32_t i = 0; i < BLOCK / 64; i++) {
        const u32 x = s_wave[i];
        if (i < w) base += x;
        all += x;
    }
    __syncthreads();
    total = all;
    return base + incl - v;
}

__global__ void __launch_bounds__(kStageBlock) k_sort_spread_staged(const uint32_t* __restrict__ d_scalars, int is_mont,
                                                                  BatchGeom bg, uint32_t table_stride, MsmConfig cfg,
                                                                  uint32_t tile, uint32_t fine_bits, uint32_t coarse_bins,
                                                                  const uint32_t* __restrict__ d_cnt, const uint32_t* __restrict__ d_binfill,
                                                                  uint32_t ch, uint32_t* __restrict__ d_binstart,
                                                                  uint32_t* __restrict__ d_prefix, uint32_t* __restrict__ d_total,
                                                                  uint32_t* __restrict__ d_pairs32) {
    __shared__ u32 s_cur[kMaxCoarse];    // the tile's cursor in every bin (global position of its next pair)
    __shared__ u32 s_cnt[kMaxCoarse];    // round: pairs per bin, then the bins' cursors inside the staging buffer
    __shared__ u32 s_gd[kMaxCoarse];     // round: global position minus staging slot, per bin
    __shared__ u32 s_scan[kStageBlock];
    __shared__ u32 s_stage[kStageSlots];
    __shared__ uint16_t s_sbin[kStageSlots];
    spread_tile_cursors<kStageBlock>(coarse_bins, d_cnt, d_binfill, ch, d_binstart, d_prefix, d_total, s_cur, s_scan);
    const uint32_t t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * tile;
    const uint64_t total = (uint64_t)bg.n * bg.batch;
    const u32 fine_mask = (1u << fine_bits) - 1u;
    const uint32_t per_round = kStageSlots / cfg.max_digits < (uint32_t)kStageBlock ? kStageSlots / cfg.max_digits : (uint32_t)kStageBlock;
    constexpr uint32_t kBinsPerLane = kMaxCoarse / kStageBlock;  // lane t owns the bins [t * kBinsPerLane, ...) in the scan
    for (uint32_t r0 = 0; r0 < tile; r0 += per_round) {
        for (uint32_t q = t; q < coarse_bins; q += kStageBlock) s_cnt[q] = 0;
        __syncthreads();
        const uint32_t off = r0 + t;
        const uint64_t g = base + off;
        const bool valid = t < per_round && off < tile && g < total;
        u32 k[8];
        bool flip = false;
        uint32_t i = 0;
        u32 pb = 0;
        if (valid) {
            const uint32_t p = (uint32_t)(g / bg.n);
            i = (uint32_t)(g - (uint64_t)p * bg.n);
            pb = p * bg.nb;
            flip = load_scalar(d_scalars, p * bg.stride + i, is_mont, k);
            u32 k1[8];
#pragma unroll
            for (int w = 0; w < 8; w++) k1[w] = k[w];
            for_each_digit(k1, cfg, [&](uint32_t, u32 bkt, bool) { atomicAdd(&s_cnt[(pb + bkt) >> fine_bits], 1u); });
        }
        __syncthreads();
        // scan of the round's counts: lane t sums its kBinsPerLane consecutive bins, the sums are scanned, then every bin gets
        // its first staging slot; the tile's cursor advances by the round's pairs
        u32 mine[kBinsPerLane], sum = 0;
#pragma unroll
        for (uint32_t e = 0; e < kBinsPerLane; e++) {
            const uint32_t q = t * kBinsPerLane + e;
            mine[e] = q < coarse_bins ? s_cnt[q] : 0u;
            sum += mine[e];
        }
        u32 round_total;
        u32 run = block_exclusive_scan_fast<kStageBlock>(sum, s_scan, round_total);
#pragma unroll
        for (uint32_t e = 0; e < kBinsPerLane; e++) {
            const uint32_t q = t * kBinsPerLane + e;
            if (q < coarse_bins) {
                const u32 cur = s_cur[q];
                s_gd[q] = cur - run;
                s_cur[q] = cur + mine[e];
                s_cnt[q] = run;
            }
            run += mine[e];
        }
        __syncthreads();
        if (valid) {
            for_each_digit(k, cfg, [&](uint32_t j, u32 bkt, bool neg) {
                const u32 b = pb + bkt, bin = b >> fine_bits;
                const u32 slot = atomicAdd(&s_cnt[bin], 1u);
                s_stage[slot] = (j * table_stride + i) | ((neg != flip) ? 0x80000000u : 0u) | ((b & fine_mask) << kPackedIndexBits);
                s_sbin[slot] = (uint16_t)bin;
            });
        }
        __syncthreads();
        for (uint32_t e = t; e < round_total; e += kStageBlock) d_pairs32[s_gd[s_sbin[e]] + e] = s_stage[e];
        // (no barrier here: the next round's first barrier stands between these reads and its writes of s_gd / s_stage / s_sbin;
        // s_cnt, which it clears before that barrier, is not read above)
    }
}

// ---- fine passes, tiled: every coarse bin [rs, re) of d_pairs is cut into chunks of `ch` pairs, one workgroup per
// chunk, so that a bin holding most of the references (a 0/1 polynomial puts ALL of them into one bucket) is
// sorted by as many workgroups as a uniform input uses.
//   count    chunk (bin, j): LDS histogram of the fine key -> H[prefix[bin]*fine + key*nch(bin) + j]
//   order    inside a bin the final order is key-major, chunk-minor: exactly the memory order of the bin's H entries,
//            so a position = bin start + exclusive prefix over the bin's H.  Ordinary bins (<= kFineLocal chunks)
//            leave that to their scatter workgroups; k_fine_binscan does it for the others
//   scatter  chunk (bin, j): LDS cursors from H, references to their final position

// pairs per chunk: at most 2048 chunks come from the length, at most coarse_bins (<= 2048) from rounding up per bin
static uint32_t fine_chunk_len(uint64_t max_pairs) {
    static const uint64_t target = [] {
        const char* v = std::getenv("KZG_FINE_CHUNKS");
        uint64_t t = v ? std::strtoull(v, nullptr, 10) : 2048ull;
        return t < 1 ? 1ull : (t > 2048 ? 2048ull : t);
    }();
    uint64_t ch = (max_pairs + target - 1) / target;
    ch = (ch + 255) / 256 * 256;
    return (uint32_t)(ch < 4096 ? 4096 : ch);
}

// chunk k -> (bin, first pair, last pair, chunks of the bin, index inside the bin); false when k is past the end
struct FineChunk {
    uint32_t bin, beg, end, nch, j, base;
};
KZG_DEV bool locate_chunk(uint32_t k, const uint32_t* __restrict__ d_prefix, const uint32_t* __restrict__ d_binstart,
                          uint32_t coarse_bins, uint32_t ch, FineChunk& c) {
    if (k >= d_prefix[coarse_bins]) return false;
    uint32_t lo = 0, hi = coarse_bins;  // smallest hi with prefix[hi] > k; prefix[lo] <= k
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (d_prefix[mid] <= k) lo = mid; else hi = mid;
    }
    c.bin = lo;
    c.base = d_prefix[lo];
    c.nch = d_prefix[lo + 1] - c.base;
    c.j = k - c.base;
    const uint32_t rs = d_binstart[lo], re = d_binstart[lo + 1];
    c.beg = rs + c.j * ch;
    c.end = (re - c.beg < ch) ? re : c.beg + ch;
    return true;
}

// a pair in either form: its fine key and the reference that goes out (sign | table index)
template <bool kPacked> struct SortPair;
template <> struct SortPair<false> {
    typedef uint64_t type;
    static KZG_DEV u32 key(uint64_t p) { return (u32)(p >> 32); }
    static KZG_DEV u32 ref(uint64_t p) { return (u32)p; }
};
template <> struct SortPair<true> {
    typedef uint32_t type;
    static KZG_DEV u32 key(uint32_t p) { return (p >> kPackedIndexBits) & ((1u << kPackedFineBits) - 1u); }
    static KZG_DEV u32 ref(uint32_t p) { return p & (0x80000000u | ((1u << kPackedIndexBits) - 1u)); }
};

template <bool kPacked>
__global__ void __launch_bounds__(kSortBlock) k_fine_count(const typename SortPair<kPacked>::type* __restrict__ d_pairs,
                                                           const uint32_t* __restrict__ d_binstart, uint32_t fine_bits,
                                                           uint32_t coarse_bins, uint32_t ch,
                                                           const uint32_t* __restrict__ d_prefix,
                                                           uint32_t* __restrict__ d_table, uint32_t* __restrict__ d_binfill) {
    __shared__ u32 s_hist[kFineMax];
    // the bin fill counters have been consumed (k_sort_spread is complete): zero again for the slot's next job
    if (blockIdx.x == 0)
        for (uint32_t q = threadIdx.x; q < coarse_bins; q += kSortBlock) d_binfill[q] = 0;
    FineChunk c;
    if (!locate_chunk(blockIdx.x, d_prefix, d_binstart, coarse_bins, ch, c)) return;
    const uint32_t fine = 1u << fine_bits;
    if (threadIdx.x < fine) s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t e = c.beg + threadIdx.x; e < c.end; e += kSortBlock) atomicAdd(&s_hist[SortPair<kPacked>::key(d_pairs[e])], 1u);
    __syncthreads();
    if (threadIdx.x < fine) d_table[(size_t)c.base * fine + (size_t)threadIdx.x * c.nch + c.j] = s_hist[threadIdx.x];
}

// One workgroup per coarse bin.  Empty bin: its buckets all start (and end) at the bin start.  A bin of more than
// kFineLocal chunks: exclusive scan of its fine * nch entries of H in place, plus the bin start -> global positions
// (thread t owns a contiguous run; skewed inputs only, e.g. a polynomial whose coefficients are all equal).
__global__ void __launch_bounds__(kSortBlock) k_fine_binscan(const uint32_t* __restrict__ d_binstart, uint32_t fine_bits,
                                                             const uint32_t* __restrict__ d_prefix,
                                                             uint32_t* __restrict__ d_table, uint32_t* __restrict__ d_offs) {
    __shared__ u32 s_scan[kSortBlock];
    const uint32_t bin = blockIdx.x, t = threadIdx.x;
    const uint32_t fine = 1u << fine_bits;
    const uint32_t base = d_prefix[bin], nch = d_prefix[bin + 1] - base;
    const uint32_t rs = d_binstart[bin];
    if (nch == 0) {
        if (t < fine) d_offs[bin * fine + t] = rs;
        return;
    }
    if (nch <= kFineLocal) return;
    u32* h = d_table + (size_t)base * fine;
    const uint32_t len = fine * nch;
    const uint32_t per = (len + kSortBlock - 1) / kSortBlock;
    const uint32_t lo = t * per < len ? t * per : len, hi = lo + per < len ? lo + per : len;
    u32 sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += h[i];
    u32 total;
    u32 run = rs + block_exclusive_scan_256(sum, s_scan, total);
    for (uint32_t i = lo; i < hi; i++) {
        const u32 v = h[i];
        h[i] = run;
        run += v;
    }
}

// Scatter of one chunk, staged through LDS: the references are first ranked inside the workgroup (one LDS cursor per
// fine key, cursors start at the key's offset inside the chunk), stored in LDS in that order together with their
// global destination, and then written out by consecutive lanes -- a key's run of this chunk is contiguous in global
// memory, so a wave's 64 stores fall into a handful of 64-byte runs instead of 64 different ones (the direct form
// issued 15.7 M separate 4-byte store transactions per commitment: 145 us alone at 2^20 terms).
constexpr uint32_t kFineStage = 4096;  // references staged per round (16 KB + 16 KB of LDS)
template <bool kPacked>
__global__ void __launch_bounds__(kSortBlock) k_fine_scatter(const typename SortPair<kPacked>::type* __restrict__ d_pairs,
                                                             const uint32_t* __restrict__ d_binstart,
                                                             uint32_t fine_bits, uint32_t coarse_bins, uint32_t ch,
                                                             const uint32_t* __restrict__ d_prefix,
                                                             const uint32_t* __restrict__ d_table,
                                                             uint32_t* __restrict__ d_offs,
                                                             uint32_t* __restrict__ d_sorted) {
    __shared__ u32 s_cur[kFineMax];    // global cursor of every fine key for this chunk
    __shared__ u32 s_cnt[kFineMax];    // round: references per key, then exclusive offsets, then local cursors
    __shared__ u32 s_ref[kFineStage];  // round: references in key order
    __shared__ u32 s_dst[kFineStage];  // round: their global positions
    FineChunk c;
    if (!locate_chunk(blockIdx.x, d_prefix, d_binstart, coarse_bins, ch, c)) return;
    const uint32_t fine = 1u << fine_bits;
    const uint32_t t = threadIdx.x;
    {
        // where this chunk's run of every key starts.  Ordinary bin: from the bin's own H rows (thread t = key t reads
        // its nch <= kFineLocal counts: all of them give the key's size, those of the chunks in front give the offset
        // inside the key's run), the keys' sizes are scanned in LDS.  Large bin: k_fine_binscan left the position.
        u32 mine = 0;
        if (c.nch <= kFineLocal) {
            u32 row = 0, front = 0;
            if (t < fine) {
                const u32* h = d_table + (size_t)c.base * fine + (size_t)t * c.nch;
                for (uint32_t i = 0; i < c.nch; i++) {
                    const u32 v = h[i];
                    row += v;
                    if (i < c.j) front += v;
                }
            }
            u32 total;
            const u32 key_start = d_binstart[c.bin] + block_exclusive_scan_256(row, s_ref, total);
            __syncthreads();
            mine = key_start + front;
            if (c.j == 0 && t < fine) d_offs[c.bin * fine + t] = key_start;
        } else if (t < fine) {
            mine = d_table[(size_t)c.base * fine + (size_t)t * c.nch + c.j];
            if (c.j == 0) d_offs[c.bin * fine + t] = mine;
        }
        if (t < fine) s_cur[t] = mine;
    }
    for (uint32_t r0 = c.beg; r0 < c.end; r0 += kFineStage) {
        const uint32_t r1 = (c.end - r0 < kFineStage) ? c.end : r0 + kFineStage;
        if (t < kFineMax) s_cnt[t] = 0;
        __syncthreads();
        // pass 1: count per key inside the round (pairs stay in registers: kFineStage / kSortBlock = 16 per lane)
        typename SortPair<kPacked>::type pr[kFineStage / kSortBlock];
#pragma unroll
        for (uint32_t q = 0; q < kFineStage / kSortBlock; q++) {
            const uint32_t e = r0 + q * kSortBlock + t;
            pr[q] = e < r1 ? d_pairs[e] : (typename SortPair<kPacked>::type)0;
            if (e < r1) atomicAdd(&s_cnt[SortPair<kPacked>::key(pr[q])], 1u);
        }
        __syncthreads();
        // exclusive scan of the (at most 256) counts: one lane per key, Hillis-Steele in LDS
        u32 mine = t < kFineMax ? s_cnt[t] : 0u;
        u32 incl = mine;
        __syncthreads();
        if (t < kFineMax) s_ref[t] = incl;  // scratch for the scan (s_ref is free until pass 2)
        __syncthreads();
        for (uint32_t off = 1; off < kFineMax; off <<= 1) {
            u32 add = (t < kFineMax && t >= off) ? s_ref[t - off] : 0u;
            __syncthreads();
            if (t < kFineMax) {
                incl += add;
                s_ref[t] = incl;
            }
            __syncthreads();
        }
        const u32 excl = incl - mine;
        // global base of this key's run for THIS round = global cursor; local cursor = exclusive offset
        u32 gbase = 0;
        if (t < kFineMax) {
            gbase = s_cur[t];
            s_cur[t] = gbase + mine;  // next round continues behind it
            s_cnt[t] = excl;
        }
        __syncthreads();
        if (t < kFineMax) s_dst[t] = gbase - excl;  // global position = (gbase - excl) + local rank; parked per key
        __syncthreads();
        // pass 2: rank inside the round, park reference and destination in key order
        u32 keep_dst[kFineStage / kSortBlock];
        u32 keep_pos[kFineStage / kSortBlock];
#pragma unroll
        for (uint32_t q = 0; q < kFineStage / kSortBlock; q++) {
            const uint32_t e = r0 + q * kSortBlock + t;
            if (e < r1) {
                const u32 key = SortPair<kPacked>::key(pr[q]);
                const u32 pos = atomicAdd(&s_cnt[key], 1u);
                keep_pos[q] = pos;
                keep_dst[q] = s_dst[key] + pos;
            }
        }
        __syncthreads();  // every lane has read s_dst[key] before it is overwritten
#pragma unroll
        for (uint32_t q = 0; q < kFineStage / kSortBlock; q++) {
            const uint32_t e = r0 + q * kSortBlock + t;
            if (e < r1) {
                s_ref[keep_pos[q]] = SortPair<kPacked>::ref(pr[q]);
                s_dst[keep_pos[q]] = keep_dst[q];
            }
        }
        __syncthreads();
        // pass 3: consecutive lanes write consecutive ranks
        const uint32_t cnt = r1 - r0;
        for (uint32_t i = t; i < cnt; i += kSortBlock) d_sorted[s_dst[i]] = s_ref[i];
        __syncthreads();
    }
}

// ---- exclusive scan of 256 values, one per thread -------------------------------------------------------------------
template <int N>
__device__ __forceinline__ u32 block_exclusive_scan_n(u32 v, u32* lds, u32& total) {
    // Hillis-Steele over N values in LDS
    int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < N; off <<= 1) {
        u32 add = t >= off ? lds[t - off] : 0u;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    total = lds[N - 1];
    return lds[t] - v;
}
__device__ __forceinline__ u32 block_exclusive_scan_256(u32 v, u32* lds, u32& total) { return block_exclusive_scan_n<256>(v, lds, total); }

// ---- small inputs: the whole sort in one workgroup ----------------------------------------------------------
// A commitment of a few thousand terms is bound by the latency of its chain of dependent kernels (the twelve
// launches above cost ~0.25 ms of the ~0.9 ms at degree 1000), so up to kSmallSortScalars scalars and kSmallSortBuckets
// buckets are handled by one workgroup: histogram in LDS, scan, scatter.
constexpr uint32_t kSmallSortScalars = 4096;
constexpr uint32_t kSmallSortBuckets = 4096;
constexpr int kSmallSortBlock = 1024;

__global__ void __launch_bounds__(kSmallSortBlock) k_sort_small(const uint32_t* __restrict__ d_scalars, int is_mont, uint32_t n,
                                                                uint32_t table_stride, MsmConfig cfg,
                                                                uint32_t* __restrict__ d_offs,
                                                                uint32_t* __restrict__ d_sorted,
                                                                uint32_t* __restrict__ d_header) {
    __shared__ u32 s_hist[kSmallSortBuckets];
    if (blockIdx.x == 0 && threadIdx.x < kHeavyHeaderBytes / 4) d_header[threadIdx.x] = 0;  // the job's counters
    __shared__ u32 s_part[kSmallSortBlock];
    const uint32_t t = threadIdx.x;
    const uint32_t nb = cfg.nb;
    for (uint32_t b = t; b < nb; b += kSmallSortBlock) s_hist[b] = 0;
    __syncthreads();
    for (uint32_t i = t; i < n; i += kSmallSortBlock) {
        u32 k[8];
        (void)load_scalar(d_scalars, i, is_mont, k);
        for_each_digit(k, cfg, [&](uint32_t, u32 bkt, bool) { atomicAdd(&s_hist[bkt], 1u); });
    }
    __syncthreads();
    // exclusive scan of nb <= 4096 counts: each lane owns nb / 1024 consecutive buckets (at least one)
    const uint32_t per = (nb + kSmallSortBlock - 1) / kSmallSortBlock;
    u32 local = 0;
    for (uint32_t q = 0; q < per; q++) {
        uint32_t b = t * per + q;
        if (b < nb) local += s_hist[b];
    }
    s_part[t] = local;
    __syncthreads();
    for (int off = 1; off < kSmallSortBlock; off <<= 1) {
        u32 add = t >= (uint32_t)off ? s_part[t - off] : 0u;
        __syncthreads();
        s_part[t] += add;
        __syncthreads();
    }
    u32 run = s_part[t] - local;
    for (uint32_t q = 0; q < per; q++) {
        uint32_t b = t * per + q;
        if (b < nb) {
            u32 c = s_hist[b];
            d_offs[b] = run;
            s_hist[b] = run;  // becomes the cursor
            run += c;
        }
    }
    if (t == kSmallSortBlock - 1) d_offs[nb] = s_part[t];
    __syncthreads();
    // Every workgroup of the launch has computed the same offsets; each scatters the references of its own range of
    // buckets (the scattered 4-byte stores of one CU are what this kernel waits for: 51 us at degree 2500 alone).
    const uint32_t b_lo = (uint32_t)((uint64_t)nb * blockIdx.x / gridDim.x), b_hi = (uint32_t)((uint64_t)nb * (blockIdx.x + 1) / gridDim.x);
    for (uint32_t i = t; i < n; i += kSmallSortBlock) {
        u32 k[8];
        const bool flip = load_scalar(d_scalars, i, is_mont, k);
        for_each_digit(k, cfg, [&](uint32_t j, u32 bkt, bool neg) {
            if (bkt >= b_lo && bkt < b_hi) {
                u32 pos = atomicAdd(&s_hist[bkt], 1u);
                d_sorted[pos] = (j * table_stride + i) | ((neg != flip) ? 0x80000000u : 0u);
            }
        });
    }
}

bool launch_bucket_sort(hipStream_t s, const uint32_t* d_scalars, int is_mont, uint32_t n, uint32_t batch,
                        uint64_t stride, uint32_t table_stride, MsmConfig cfg, uint32_t* d_cnt, uint32_t* d_ws,
                        uint64_t* d_pairs, uint32_t* d_offs, uint32_t* d_sorted, uint32_t* d_header) {
    if (n == 0 || batch == 0) return false;
    static const bool small_path = [] { const char* v = std::getenv("KZG_SMALL_SORT"); return !(v && v[0] == '0'); }();
    if (small_path && batch == 1 && n <= kSmallSortScalars && cfg.nb <= kSmallSortBuckets) {
        uint32_t groups = n / 256;  // workgroups sharing the scatter: 1 ... 8
        groups = groups < 1 ? 1 : (groups > 8 ? 8 : groups);
        hipLaunchKernelGGL(k_sort_small, dim3(groups), dim3(kSmallSortBlock), 0, s, d_scalars, is_mont, n, table_stride, cfg, d_offs,
                           d_sorted, d_header);
        return true;
    }
    const uint32_t nb_total = cfg.nb * batch;
    SortGeom g = sort_geometry((uint64_t)n * batch, nb_total, cfg, table_stride);
    BatchGeom bg{n, batch, stride, cfg.nb};
    uint32_t* d_binstart = d_ws + kWsBinStart;
    uint32_t* d_prefix = d_ws + kWsPrefix;
    uint32_t* d_table = d_ws + kWsTable;
    uint32_t* d_total = d_offs + nb_total;  // number of references, also the end of the offsets
    const uint64_t max_pairs = (uint64_t)n * batch * cfg.max_digits;
    const uint32_t ch = fine_chunk_len(max_pairs);
    const uint32_t max_chunks = (uint32_t)((max_pairs + ch - 1) / ch) + g.coarse_bins;  // <= kFineMaxChunks
    uint32_t* d_binfill = d_ws + kWsBinFill;
    hipLaunchKernelGGL(k_sort_count, dim3(g.tiles), dim3(kRecodeBlock), 0, s, d_scalars, is_mont, bg, cfg, g.tile,
                       g.fine_bits, g.coarse_bins, d_cnt, d_binfill, d_header);
    static const bool staged = [] { const char* v = std::getenv("KZG_SPREAD_STAGED"); return !(v && v[0] == '0'); }();
    if (g.packed && staged && cfg.max_digits <= kStageSlots)
        hipLaunchKernelGGL(k_sort_spread_staged, dim3(g.tiles), dim3(kStageBlock), 0, s, d_scalars, is_mont, bg, table_stride, cfg,
                           g.tile, g.fine_bits, g.coarse_bins, d_cnt, d_binfill, ch, d_binstart, d_prefix, d_total,
                           reinterpret_cast<uint32_t*>(d_pairs));
    else
        hipLaunchKernelGGL(k_sort_spread, dim3(g.tiles), dim3(kRecodeBlock), 0, s, d_scalars, is_mont, bg, table_stride, cfg,
                           g.tile, g.tiles, g.fine_bits, g.coarse_bins, d_cnt, d_binfill, ch, d_binstart, d_prefix, d_total, d_pairs,
                           g.packed ? 1 : 0);
    if (g.packed)
        hipLaunchKernelGGL(k_fine_count<true>, dim3(max_chunks), dim3(kSortBlock), 0, s, reinterpret_cast<const uint32_t*>(d_pairs),
                           d_binstart, g.fine_bits, g.coarse_bins, ch, d_prefix, d_table, d_binfill);
    else
        hipLaunchKernelGGL(k_fine_count<false>, dim3(max_chunks), dim3(kSortBlock), 0, s, d_pairs, d_binstart, g.fine_bits,
                           g.coarse_bins, ch, d_prefix, d_table, d_binfill);
    hipLaunchKernelGGL(k_fine_binscan, dim3(g.coarse_bins), dim3(kSortBlock), 0, s, d_binstart, g.fine_bits, d_prefix, d_table, d_offs);
    if (g.packed)
        hipLaunchKernelGGL(k_fine_scatter<true>, dim3(max_chunks), dim3(kSortBlock), 0, s, reinterpret_cast<const uint32_t*>(d_pairs),
                           d_binstart, g.fine_bits, g.coarse_bins, ch, d_prefix, d_table, d_offs, d_sorted);
    else
        hipLaunchKernelGGL(k_fine_scatter<false>, dim3(max_chunks), dim3(kSortBlock), 0, s, d_pairs, d_binstart, g.fine_bits,
                           g.coarse_bins, ch, d_prefix, d_table, d_offs, d_sorted);
    return true;
}

}  // namespace kzg
