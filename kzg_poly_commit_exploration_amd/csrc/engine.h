// engine.h -- internal interface between the C-ABI (api.hip) and the kernel translation units.
// Not installed; the public boundary is include/kzg_mi355x.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace kzg {

// One affine SRS / table point in HBM: x as 13 signed radix-2^30 digits in words 0..12, y in words 16..28
// (field30.hip.h: Montgomery R' = 2^390, magnitude < 0.62 p), all zero = infinity: a 128-byte record, so that every
// random gather of the accumulation kernel touches exactly one 128-byte line.  Table layout:
// level-major, T[j * n + i] = 2^(level_bits*j) * SRS[i]  (j < W), so level 0 is the SRS itself.
constexpr size_t kAffineBytes = 128;
constexpr size_t kAffineU4 = kAffineBytes / 16;  // record stride in uint4 units
// One XYZZ accumulator in HBM: X, Y, ZZ, ZZZ as four groups of 13 signed radix-2^30 digits (field30.hip.h,
// Montgomery R' = 2^390, lazily reduced), each group padded to 16 words: 256 bytes, all zero = infinity.
constexpr size_t kXyzzBytes = 256;
constexpr size_t kXyzzU4 = kXyzzBytes / 16;      // record stride in uint4 units
constexpr size_t kXyzzWords64 = kXyzzBytes / 8;  // record stride in u64 units (host side)

// Two scalar recodings share every kernel after the digit loop:
//   kRecodeWindows  aligned signed windows of c bits: W = ceil(255/c) digits per scalar, one table level per
//                   window (level_bits = c), digit magnitude 1 .. 2^(c-1) -> 2^(c-1) buckets of weight b+1.
//   kRecodeNaf      width-c non-adjacent form: digits are odd, |d| < 2^(c-1), at most one per c consecutive
//                   bits and on average one per c+1 bits (vs one per c-1 for the same bucket count with
//                   windows: ~11 % fewer mixed additions), but a digit can start at ANY bit, so the table
//                   carries one level per scalar bit (W = 255, level_bits = 1): 255 x n x 128 B, 34 GB at
//                   2^20 points, which the 288 GB of HBM3E hold easily.  2^(c-2) buckets of weight 2b+1.
//                   Opt-in (KZG_MSM_RECODE=naf): bit-exact, but on MI355X the fewer additions are eaten by
//                   address-translation misses of the gathers over the larger table (msm_sort.hip).
enum : uint32_t { kRecodeWindows = 0, kRecodeNaf = 1 };
struct MsmConfig {
    uint32_t recode;      // kRecodeWindows | kRecodeNaf
    uint32_t c;           // digit width in bits
    uint32_t W;           // table levels
    uint32_t level_bits;  // doublings between consecutive table levels
    uint32_t nb;          // buckets per polynomial (power of two)
    uint32_t max_digits;  // bound on the non-zero digits of one scalar
};

// table_budget_bytes: what the table may occupy (the NAF recoding is refused when its table does not fit)
MsmConfig choose_msm_config(size_t n_points, size_t table_budget_bytes);

// Segment length of the bucket accumulation for M sorted references on `lanes` lanes (msm_accum.hip): computed
// on the device from the actual M, so that scalars with many zero digits still fill every lane.
// shortest segment: 8 references, or 4 for tiny jobs (up to kTinyRefs references: latency-bound, every dependent
// addition counts; measured at degree 1000: 0.77 -> 0.67 ms, while at 16384 terms the extra partials cost 2x)
constexpr uint32_t kTinyRefs = 65536;
__host__ __device__ inline uint32_t accumulate_min_seg(uint64_t refs) { return refs <= kTinyRefs ? 4u : 8u; }
__host__ __device__ inline uint32_t accumulate_seg_len(uint32_t M, uint32_t lanes) {
    uint32_t L = (M + lanes - 1) / lanes;
    const uint32_t lo = accumulate_min_seg(M);
    return L < lo ? lo : L;
}

// ---- msm_kernels.hip --------------------------------------------------------------------
constexpr size_t kHeavyHeaderBytes = 1024;  // zeroed per job: long-bucket counters (line 0), phase counters of the one-launch paths (own lines)
// scalar recoding + two-level LDS counting sort of `batch` polynomials of n terms at once (polynomial p
// at d_scalars + p * stride scalars; its buckets are [p * nb, (p+1) * nb)): fills
// d_offs[0 .. batch*nb] (last = number of references) and d_sorted (bucket-major table references,
// index | sign << 31).  d_cnt: sort_count_entries(max_batch, cfg) u32; d_ws: sort_workspace_words() u32;
// d_pairs: batch * n * max_digits u64.  batch <= sort_max_batch(cfg).
uint32_t sort_count_entries(uint32_t max_batch, MsmConfig cfg);
uint32_t sort_max_batch(MsmConfig cfg);
uint32_t sort_workspace_words();
uint32_t sort_workspace_zero_words();  // leading words of d_ws that must be zero when the workspace is first used (the sort keeps them so)
// d_header: kHeavyHeaderBytes that the job wants zeroed before its next kernel; returns true when the sort did that
// itself (it does whenever it launches anything), false when the caller has to memset them (n == 0).
bool launch_bucket_sort(hipStream_t s, const uint32_t* d_scalars, int scalars_are_mont, uint32_t n, uint32_t batch,
                        uint64_t stride, uint32_t table_stride, MsmConfig cfg, uint32_t* d_cnt,
                        uint32_t* d_ws, uint64_t* d_pairs, uint32_t* d_offs, uint32_t* d_sorted, uint32_t* d_header);
// bucket accumulation (dominant kernel): one lane per segment of L sorted references
constexpr uint32_t kMaxAccumLanes = 262144 + 64;  // bound on accumulate_lanes()
// lanes (= segments) for at most max_refs references; a multiple of the workgroup size
// alone: no other job is in flight on the context (the light kernels of other slots need no room)
uint32_t accumulate_lanes(uint64_t max_refs, bool alone = false);
// d_pair_scratch: accumulate_pair_scratch_bytes(max_refs) of scratch for the affine front end (prefix products of the
// shared inversions), or null to run plain mixed additions; max_refs bounds the references of this launch
size_t accumulate_pair_scratch_bytes(uint64_t max_refs);
void launch_bucket_accumulate(hipStream_t s, const void* d_table, const uint32_t* d_sorted, const uint32_t* d_offs,
                              uint32_t nb, uint32_t lanes, void* d_buckets /* complete buckets only: the rest by the finalisation */,
                              void* d_part_a, void* d_part_b, uint32_t lds_reserve_bytes, void* d_pair_scratch,
                              uint64_t max_refs, void* d_clock = nullptr /* kAccumClockOffset into the job's zeroed header, or null */);
// the accumulation kernel's own start / end stamps: two u64 in the job header (zeroed by the sort), copied next to the
// reference count by the finalisation (d_refs_out[2..5])
constexpr size_t kAccumClockOffset = 768;  // bytes into d_heavy_ws (words 192..195 of the header: unused by the counters)
// adds the head / tail partials of buckets that span several segments (serial for short runs, three passes of
// 64-wide trees for long ones) and writes empty buckets as infinity: every bucket is written once per job, the array is
// never cleared.  d_heavy_ws: heavy_workspace_bytes() of scratch whose first kHeavyHeaderBytes (the counters) are zero
// (launch_bucket_sort does that) -- ahead of time, so that nothing sits between the end of the accumulation and this
// launch (a fill kernel there lets the next slot's accumulation take the chip first: +0.9 ms)
size_t heavy_workspace_bytes();
// group: quads per bucket (finalize_group_size(nb); 1 = one quad per bucket, the throughput form).
void launch_bucket_finalize(hipStream_t s, const uint32_t* d_offs, uint32_t nb, uint32_t lanes, const void* d_part_a,
                            const void* d_part_b, void* d_buckets, void* d_heavy_ws,
                            uint32_t* d_refs_out /* receives the number of references, may be null */, uint32_t group = 1);
uint32_t finalize_group_size(uint32_t nb);
// Small jobs (at most kTinyRefs references, one polynomial): accumulation, finalisation, long-bucket trees and both
// reduction stages in ONE launch (msm_finalize.hip: k_small_msm).  The first kHeavyHeaderBytes of d_heavy_ws must be zero.
// lds_bytes: small_msm_lds_bytes() once hipFuncAttributeMaxDynamicSharedMemorySize has been raised for
// small_msm_kernel(), otherwise at most 64 KiB.
struct TreeSumDesc;
uint32_t small_msm_lds_bytes();
const void* small_msm_kernel();
void launch_small_msm(hipStream_t s, const void* d_table, const uint32_t* d_sorted, const uint32_t* d_offs, uint32_t nb,
                      uint32_t lanes, uint64_t max_refs, void* d_buckets, void* d_part_a, void* d_part_b, void* d_heavy_ws,
                      uint32_t* d_refs_out, const TreeSumDesc* stage1 /* 2 */, const TreeSumDesc* stage2 /* 4 */,
                      uint32_t lds_bytes);
// out[g] = sum_{q<len} in[g*gstride + q*estride], XYZZ records: log-depth tree per group;
// up to four independent jobs per launch
// group g reads in[(g / inner) * ostride + (g % inner) * gstride + q * estride], q < len
struct TreeSumDesc {
    const void* in;
    void* out;
    uint32_t groups, len;
    uint64_t gstride, estride;
    uint32_t inner;    // groups per outer block (= groups when there is no batch dimension)
    uint64_t ostride;  // record stride between outer blocks
};
void launch_tree_sums(hipStream_t s, const TreeSumDesc* descs, uint32_t count, bool dense = false, bool alone = true);
// stage2 reads what stage1 wrote.  One launch when both stages fit the chip at once (d_sync: two zeroed words, e.g.
// words 4 and 5 of d_heavy_ws), otherwise two launches.
void launch_tree_sums_two_stage(hipStream_t s, const TreeSumDesc* stage1, uint32_t count1, const TreeSumDesc* stage2,
                                uint32_t count2, uint32_t* d_sync, bool alone);

// ---- srs_kernels.hip --------------------------------------------------------------------
// blst_p1 Jacobian (host layout, strided) already copied to d_jac (n x 144 B contiguous) -> affine
void launch_jacobian_to_affine(hipStream_t s, const void* d_jac, uint32_t n, void* d_affine_out,
                               void* d_prefix_tmp);
// table level j from level j-1:  T[j][i] = 2^c * T[j-1][i]
void launch_table_window(hipStream_t s, const void* d_prev_affine, uint32_t n, uint32_t c,
                         void* d_xyzz_tmp, void* d_prefix_tmp, void* d_next_affine);
// fixed-base trusted setup: out[i] = [s^(first+i)] G1, affine
void launch_srs_generate(hipStream_t s, const uint32_t* secret_raw8 /* 256-bit LE integer */, uint64_t first, uint32_t n,
                         void* d_gtable, void* d_xyzz_tmp, void* d_prefix_tmp, void* d_affine_out);
size_t srs_gtable_bytes();
// ---- srs_io.hip ------------------------------------------------------------------------
// n x 96-byte affine points (x, y as blst_fp; (0, 0) = infinity) -> table level 0
void launch_affine96_to_table(hipStream_t s, const void* d_affine96, uint32_t n, void* d_table);
// n x 48-byte compressed points (ZCash encoding) -> table level 0; *d_status (pre-set to 0xffffffff) receives
// index + 1 of the first malformed point
void launch_uncompress(hipStream_t s, const void* d_compressed, uint32_t n, void* d_table, uint32_t* d_status);

// ---- msm_accum.hip (table format) ---------------------------------------------------------
// native table entries -> blst_p1 (Z = Montgomery one / all zero for infinity)
void launch_affine_to_p1(hipStream_t s, const void* d_affine, uint32_t n, void* d_p1_out);

// ---- multi.hip: a context spanning several devices (SRS-range slices, RCCL exchange of the partials) ------------
}  // namespace kzg
#include <string>
struct kzg_ctx;
namespace kzg {
// ---- api.hip: single-device pieces the multi-device context drives (never exported) ------------------------------
// A kid of a range-split context returns its partial sums UN-normalised (Jacobian X*ZZ, Y*ZZZ, ZZ of the XYZZ total:
// two products instead of an inversion); the parent normalises once after the K-1 additions.
void ctx_set_raw_partials(kzg_ctx* ctx, bool raw);
// Host-pointer batches on ONE device: polynomial i (i < count) is the caller's polynomial first + i * step, its n
// coefficients at coeffs + (first + i * step) * stride_coeffs blst_fr values, its result at out_p1s + 18 * (first + i *
// step) (and statuses[first + i * step]).  Sub-batches flow through the context's stream slots so that the upload of
// one overlaps the kernels of the previous ones.
int ctx_commit_batch_host(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, size_t stride_coeffs, size_t first, size_t step,
                          size_t count, uint64_t* out_p1s);
int ctx_open_batch_host(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, size_t stride_coeffs, size_t first, size_t step,
                        size_t count, const uint64_t* zs, const uint64_t* ys, uint64_t* out_p1s, int* statuses);
// One device's share of a range-sharded opening (multi.hip): the slice is uploaded ONCE.  begin: slice -> the slot's
// staging buffer, H = sum_i slice[i] z^i (the quotient scan without output); finish: the carry becomes coefficient
// `len` of the staged slice, which is opened at z with claimed value `start`.  The slot stays reserved in between;
// ctx_open_slice_abort releases it when the recurrence on the host refuses the opening.
int ctx_open_slice_begin(kzg_ctx* ctx, const uint64_t* slice, size_t len, const uint64_t z[4], uint64_t out_h[4], int* slot_out);
int ctx_open_slice_finish(kzg_ctx* ctx, int slot, size_t len, const uint64_t carry[4], const uint64_t z[4],
                          const uint64_t start[4], uint64_t out_p1[18]);
void ctx_open_slice_abort(kzg_ctx* ctx, int slot);

struct MultiState;
enum : uint32_t { kMultiRange = 0, kMultiReplicate = 1 };
int multi_create(const int* devices, int ndev, uint32_t mode, MultiState** out, std::string& err);
void multi_destroy(MultiState* m);
size_t multi_srs_len(const MultiState* m);
int multi_num_devices(const MultiState* m);
uint64_t multi_rccl_exchanges(const MultiState* m);
kzg_ctx* multi_kid(MultiState* m, int g);
const char* multi_last_error(const MultiState* m);
int multi_srs_generate(MultiState* m, const uint8_t secret_be[32], uint64_t first, size_t n);
int multi_srs_load(MultiState* m, const void* first_g1, size_t stride, size_t n);
int multi_srs_load_affine(MultiState* m, const void* affine_xy, size_t n);
int multi_srs_load_compressed(MultiState* m, const uint8_t* compressed, size_t n, size_t* bad_index);
int multi_srs_read(MultiState* m, size_t index, size_t count, uint64_t* out_p1);
int multi_commit(MultiState* m, const void* scalars, int is_mont, size_t n, uint64_t out_p1[18]);
int multi_open(MultiState* m, const uint64_t* coeffs, size_t n, const uint64_t z[4], const uint64_t y[4], uint64_t out_p1[18]);
int multi_commit_batch(MultiState* m, const uint64_t* coeffs, size_t n, size_t batch, size_t stride_coeffs, uint64_t* out_p1s);
int multi_open_batch(MultiState* m, const uint64_t* coeffs, size_t n, size_t batch, size_t stride_coeffs, const uint64_t* zs,
                     const uint64_t* ys, uint64_t* out_p1s, int* statuses);
int multi_set_max_batch(MultiState* m, size_t max_batch);
uint32_t multi_mode(const MultiState* m);

// ---- poly_kernels.hip -------------------------------------------------------------------
struct PolyScratch {
    uint32_t* d_chunk;   // per-thread chunk values / carries (Fr)
    uint32_t* d_block;   // per-block aggregates (Fr)
    uint32_t* d_flags;   // the job's flag words, zero at launch: [0] = any non-zero coefficient with index >= 1, [16..23] receive c[0]
    uint32_t* d_result;  // P(z) (8 words)
};
// raises the dynamic-LDS limit of the scan kernels on the current device (kzg_ctx_create); false when refused
bool poly_prepare_device();
size_t poly_chunk_words(uint32_t n);
size_t poly_block_words(uint32_t n);
// suffix Horner scan S[i] = sum_{k>=i} c[k] z^(k-i):  q[i-1] = S[i] (i >= 1) when d_q != nullptr,
// P(z) = S[0] to scratch.d_result; flags as above.  z in Montgomery form (8 words, host copy).
// n <= 4096: one launch that also fills the slot's 64 flag words completely (P(z) at +8, c[0] at +16, flag at +0,
// zeros elsewhere); returns false (nothing enqueued) for larger n.
bool launch_quotient_single(hipStream_t s, const uint32_t* d_coeffs, uint32_t n, const uint32_t z_mont[8], uint32_t* d_q,
                            uint32_t* d_small);
void launch_quotient(hipStream_t s, const uint32_t* d_coeffs, uint32_t n, const uint32_t z_mont[8],
                     uint32_t* d_q, PolyScratch scratch);

}  // namespace kzg
