/* kzg_mi355x.h -- C-ABI of libkzg_mi355x.so: the MI355X (gfx950) engine behind the hot path of
 * VGLoic/kzg-poly-commit-exploration, i.e. Polynomial::commit and Evaluation::generate_proof.
 *
 * The reference has no FFI seam of its own: commit / generate_proof are inherent Rust methods that
 * descend into blst one scalar multiplication at a time (reference src/polynomial.rs:200-215,
 * 260-269 -> src/curves.rs:79-96).  The boundary is therefore the BODY of those two methods; every
 * entry point below cites the reference code it replaces, and INTEGRATION.md shows the Rust
 * `extern "C"` block and the two method bodies a maintainer would write against this header.
 *
 * Layouts are blst's, so Rust passes its own memory without conversion:
 *   Fr  ("blst_fr",  reference src/scalar.rs:7-8)   : 4 x uint64 little-endian limbs, Montgomery, R = 2^256
 *   Fp  ("blst_fp")                                 : 6 x uint64 little-endian limbs, Montgomery, R = 2^384
 *   G1  ("blst_p1",  reference src/curves.rs:10-17) : {x, y, z} Jacobian, 18 x uint64; z == 0 <=> infinity
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer; the library copies what
 * it keeps and never retains a host pointer past return; every function returns 0 (KZG_OK) or a
 * negative kzg_status; nothing throws or longjmps across the boundary.  A kzg_ctx drives one GPU
 * (kzg_ctx_create; one process per GPU with the RCCL exchange in the host program, see bench.py) or several GPUs of
 * one node from one process (kzg_ctx_create_multi / _ex: SRS split by point range with an RCCL exchange of the partial
 * sums inside the library, or SRS replicated with batches split by polynomial).  Every entry point may be called
 * from several threads on the same context: the synchronous host-pointer calls (kzg_commit, kzg_open, kzg_*_batch)
 * each take one of the context's stream slots and run side by side; different contexts are independent.
 *
 * There is no CPU fallback: without a HIP device kzg_ctx_create fails with KZG_ERR_NO_DEVICE.
 */
#ifndef KZG_MI355X_H
#define KZG_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kzg_ctx kzg_ctx;

/* Bumped whenever a struct or a signature below changes incompatibly (2: kzg_kernel_times lost `scan_ms` in round 2;
 * 3: kzg_ctx_create_multi_ex, host-pointer batches; 4: kzg_kernel_times gained accumulate_events_ms at its end).  kzg_abi_version() returns the value the library was built
 * with: a caller compiled against another value must not read kzg_kernel_times. */
#define KZG_ABI_VERSION 4
int kzg_abi_version(void);

typedef enum kzg_status {
    KZG_OK = 0,
    /* "Setup does not allow for commitment generation of the polynomial. The polynomial degree is
     * too high."  (reference src/polynomial.rs:201-205) */
    KZG_ERR_DEGREE_TOO_HIGH = -1,
    /* "Unable to divide a constant polynomial"  (reference src/polynomial.rs:159-167) */
    KZG_ERR_CONSTANT_POLY = -2,
    /* "[divide_by_root] Fail to divide the polynomial by a root, constant terms do not add up"
     * (reference src/polynomial.rs:184-192) */
    KZG_ERR_REMAINDER = -3,
    KZG_ERR_INVALID_ARG = -4,
    KZG_ERR_NO_DEVICE = -5,  /* no usable HIP device: the library has no CPU path */
    KZG_ERR_HIP = -6,        /* a HIP runtime call failed; kzg_last_error() has the text */
    KZG_ERR_NO_SRS = -7,     /* commit/open before kzg_srs_load_g1 / kzg_srs_generate_g1 */
    KZG_ERR_BUSY = -8        /* async slot still in flight */
} kzg_status;

/* ---- context ------------------------------------------------------------------------------ */

/* Creates the engine on HIP device `device` (streams, workspaces are sized at SRS load).
 * No reference analogue: the reference is stateless and receives the SRS slice on every call
 * (src/polynomial.rs:200); the context exists to keep the SRS resident in HBM. */
int kzg_ctx_create(int device, kzg_ctx** out);
/* One context over `ndev` HIP devices of this node (the signature SURVEY.md section 8(b) sketched).  The SRS is
 * split by point range: devices[g] keeps points [g * ceil(n/ndev), ...) resident with their window tables
 * (kzg_srs_load_g1 / kzg_srs_generate_g1 on the context do the split).  kzg_commit, kzg_commit_le_bytes and kzg_open
 * then shard transparently -- one partial MSM per device driven by its own host thread; a range-sharded opening
 * evaluates the slices, runs the ndev-step carry recurrence on the host and opens every slice extended by its carry
 * (replaces the same bodies, reference src/polynomial.rs:200-215 and 260-269) -- and the 144-byte partial sums are
 * exchanged with ncclAllGather (librccl, one communicator per device, single process) and added with kzg_g1_sum:
 * RCCL has no reduction operator for curve points, so "reduce" = all-gather + K-1 complete additions.
 * A device may be listed more than once (virtual slices: how a one-GPU box rehearses the path); such a context
 * gathers on the host, since a communicator needs distinct devices (the same fallback is taken when RCCL cannot
 * form the communicator).  kzg_evaluate / kzg_quotient run on devices[0].  The host-pointer batches
 * (kzg_commit_batch / kzg_open_batch) work on every kind of context.  The asynchronous and device-pointer entry points
 * (kzg_*_submit / kzg_wait* / kzg_dev_*) name ONE device's memory and slots: they return KZG_ERR_INVALID_ARG on a
 * multi-device context, also when ndev == 1.
 *
 * kzg_ctx_create_multi_ex(..., KZG_MULTI_REPLICATE_SRS, ...): every device keeps the WHOLE SRS and work is split by
 * polynomial instead (BASELINE config 5: 64 openings of degree 2^20 on 8 GPUs): polynomial p
 * of a kzg_commit_batch / kzg_open_batch goes to devices[p mod ndev], each device pipelines its share through its
 * stream slots, nothing is exchanged.  Single kzg_commit / kzg_open calls take the devices in turn, so caller
 * threads spread over the GPUs.  flags == 0 is kzg_ctx_create_multi. */
#define KZG_MULTI_REPLICATE_SRS 1u
int kzg_ctx_create_multi(const int* devices, int ndev, kzg_ctx** out);
int kzg_ctx_create_multi_ex(const int* devices, int ndev, unsigned flags, kzg_ctx** out);
/* devices of the context (1 for kzg_ctx_create) and how many partial-sum exchanges went through RCCL so far */
int kzg_num_devices(const kzg_ctx* ctx);
uint64_t kzg_rccl_exchanges(const kzg_ctx* ctx);
void kzg_ctx_destroy(kzg_ctx* ctx);
const char* kzg_strerror(int status);
/* text of the last KZG_ERR_HIP on this context (valid until the next call on it) */
const char* kzg_last_error(const kzg_ctx* ctx);

/* ---- SRS: the G1 half of the reference's Vec<SetupArtifact> ------------------------------- */

/* Ingests n blst_p1 values starting at first_g1 with a byte stride (= size_of::<SetupArtifact>(),
 * reference src/trusted_setup.rs:31-35; the shim passes &srs[0].g1 so no field offset is
 * assumed).  Points are Jacobian with arbitrary Z as blst_p1_mult leaves them
 * (src/trusted_setup.rs:54-62); the library normalises them to affine on the device and builds
 * its window tables.  A rank of a sharded MSM loads only its slice of the SRS and commits the
 * matching coefficient slice (indices are slice-relative). */
int kzg_srs_load_g1(kzg_ctx* ctx, const void* first_g1, size_t stride_bytes, size_t n);

/* Trusted setup on the device, G1 side only: SRS[i] = [s^i mod r]G1 for i in [first, first+n),
 * s = secret read big-endian (reference src/trusted_setup.rs:20-28, 40-62).  Next-row component
 * (SURVEY.md section 8f-2); it also makes the large bench configurations set up in seconds. */
int kzg_srs_generate_g1(kzg_ctx* ctx, const uint8_t secret_be[32], uint64_t first, size_t n);

/* Wire / on-disk forms of the SRS (all of them split by range on a multi-device context):
 *  kzg_srs_load_affine      n x 96 bytes: x, y as blst_fp (Montgomery), (0, 0) = infinity -- blst_p1_affine[n].
 *                           No normalisation pass: the points go straight into the window-table builder.
 *  kzg_srs_load_compressed  n x 48 bytes, ZCash encoding: what `Serialize for G1Point` writes into the CLI's
 *                           setup.json (reference src/curves.rs:99-110) and `Deserialize` reads back with one
 *                           blst_p1_uncompress per point (src/curves.rs:112-183).  Decompressed on the device (one
 *                           lane per point: y = (x^3 + 4)^((p+1)/4), sign from the encoding); same acceptance as
 *                           blst_p1_uncompress (compressed flag, x < p, on the curve; no subgroup check).  On a
 *                           malformed point: KZG_ERR_INVALID_ARG and *bad_index = its index (bad_index may be NULL).
 *  kzg_srs_save / kzg_srs_load_file   binary cache of the resident SRS: 128-byte header ("KZGSRS1", n, compressed
 *                           first and last point as a fingerprint of the content) + n x 96-byte affine points; loading
 *                           checks the fingerprint and then takes the kzg_srs_load_affine path. */
int kzg_srs_load_affine(kzg_ctx* ctx, const void* affine_xy, size_t n);
int kzg_srs_load_compressed(kzg_ctx* ctx, const uint8_t* compressed, size_t n, size_t* bad_index);
int kzg_srs_save(kzg_ctx* ctx, const char* path);
int kzg_srs_load_file(kzg_ctx* ctx, const char* path);

/* Copies SRS entries [index, index+count) back as blst_p1 with Z = 1 (affine), e.g. to hand them
 * to the reference's serde or to check them against blst. */
int kzg_srs_read_g1(kzg_ctx* ctx, size_t index, size_t count, uint64_t* out_p1);
size_t kzg_srs_len(const kzg_ctx* ctx);

/* ---- the hot path ------------------------------------------------------------------------- */

/* Polynomial::commit (reference src/polynomial.rs:200-215): out = sum_{i<n} coeffs[i] * SRS[i].
 * coeffs = self.coefficients.as_ptr() (n x blst_fr, Montgomery).  n == 0 -> infinity.
 * n > kzg_srs_len -> KZG_ERR_DEGREE_TOO_HIGH.  out_p1 is a blst_p1 with Z = 1 (Montgomery one) or
 * all-zero for infinity, so G1Point::from(blst_p1) (src/curves.rs:13-17) wraps it directly. */
int kzg_commit(kzg_ctx* ctx, const uint64_t* coeffs_fr_mont, size_t n, uint64_t out_p1[18]);

/* Same with scalars as n x 32 canonical little-endian bytes (< r), i.e. what Scalar::to_le_bytes
 * yields (reference src/scalar.rs:83-93) and what G1Point::mult feeds blst (src/curves.rs:93). */
int kzg_commit_le_bytes(kzg_ctx* ctx, const uint8_t* scalars_le, size_t n, uint64_t out_p1[18]);

/* Evaluation::generate_proof (reference src/polynomial.rs:260-269) fused on the device:
 * (P - y) (:128-145) / (x - z) (:150-195) then commit.  z = evaluation.point, y = evaluation.result
 * (4 x uint64 Montgomery each).  Error behaviour of the reference is reproduced:
 *   n == 0 and y == 0 -> infinity;  constant polynomial: c0 == y -> infinity, else
 *   KZG_ERR_CONSTANT_POLY;  P(z) != y -> KZG_ERR_REMAINDER;  quotient longer than the SRS ->
 *   KZG_ERR_DEGREE_TOO_HIGH. */
int kzg_open(kzg_ctx* ctx, const uint64_t* coeffs_fr_mont, size_t n, const uint64_t z[4],
             const uint64_t y[4], uint64_t out_p1[18]);

/* `batch` calls of Polynomial::commit / Evaluation::generate_proof against the same SRS in one call, HOST pointers:
 * polynomial p = n blst_fr values at coeffs + p * stride_coeffs * 4 (stride_coeffs >= n), result p at out_p1s + 18 * p.
 * This is the loop of the reference's callers (src/lib.rs:16-33 per polynomial; benches/evaluation_proof.rs:51-54)
 * handed over whole, so that uploads overlap kernels and, on a multi-device context, the polynomials spread over
 * the GPUs (replicated SRS: by polynomial, no communication; range-split SRS: every polynomial sharded, one
 * exchange per batch).  n <= kzg_srs_len (commit) / n - 1 <= kzg_srs_len (open): batches take truncated polynomials.
 * kzg_open_batch opens polynomial p at zs[p] with claimed value ys[p] and fills statuses[p] with KZG_OK,
 * KZG_ERR_CONSTANT_POLY or KZG_ERR_REMAINDER (out_p1s[p] is written only for KZG_OK); its return value reports
 * failures of the call as a whole.  kzg_set_max_batch bounds the polynomials per pass of the kernels (default 1). */
int kzg_commit_batch(kzg_ctx* ctx, const uint64_t* coeffs_fr_mont, size_t n, size_t batch, size_t stride_coeffs,
                     uint64_t* out_p1s /* batch x 18 */);
int kzg_open_batch(kzg_ctx* ctx, const uint64_t* coeffs_fr_mont, size_t n, size_t batch, size_t stride_coeffs,
                   const uint64_t* zs, const uint64_t* ys, uint64_t* out_p1s /* batch x 18 */, int* statuses);

/* Polynomial::sub + divide_by_root alone (reference src/polynomial.rs:128-195): writes the
 * quotient coefficients (Montgomery) to out_q (room for n-1 entries) and their count, after the
 * reference's trailing-zero truncation, to *out_qn.  Same error codes as kzg_open. */
int kzg_quotient(kzg_ctx* ctx, const uint64_t* coeffs_fr_mont, size_t n, const uint64_t z[4],
                 const uint64_t y[4], uint64_t* out_q, size_t* out_qn);

/* Polynomial::evaluate (reference src/polynomial.rs:112-123): y = P(z), the same device scan as the
 * quotient (y is its remainder).  Next-row component. */
int kzg_evaluate(kzg_ctx* ctx, const uint64_t* coeffs_fr_mont, size_t n, const uint64_t z[4],
                 uint64_t out_y[4]);

/* ---- device-resident / pipelined variants -------------------------------------------------
 * d_coeffs is a DEVICE pointer (n x blst_fr, Montgomery) on the context's GPU, e.g. a tensor
 * produced upstream.  submit enqueues on one of kzg_num_slots() internal HIP streams and returns
 * at once; wait blocks on that slot, finishes the tail on the host and writes the result.  Several
 * slots in flight keep the GPU busy across the latency-bound end of each MSM. */
int kzg_num_slots(const kzg_ctx* ctx);
int kzg_commit_submit(kzg_ctx* ctx, int slot, const void* d_coeffs, size_t n);
int kzg_open_submit(kzg_ctx* ctx, int slot, const void* d_coeffs, size_t n, const uint64_t z[4],
                    const uint64_t y[4]);
int kzg_wait(kzg_ctx* ctx, int slot, uint64_t out_p1[18]);

/* Batches: `batch` polynomials of n coefficients each (polynomial p at d_coeffs + p * stride_coeffs
 * blst_fr values), committed against the same SRS in ONE pass of the kernels -- the shape of
 * BASELINE config 5 and what keeps small per-GPU shards efficient when a commitment is sharded over
 * several GPUs.  kzg_set_max_batch sizes the workspaces (default 1; it is clamped to what the sort's
 * bin table allows, read it back with kzg_max_batch).  n <= kzg_srs_len. */
int kzg_set_max_batch(kzg_ctx* ctx, size_t max_batch);
size_t kzg_max_batch(const kzg_ctx* ctx);
int kzg_commit_batch_submit(kzg_ctx* ctx, int slot, const void* d_coeffs, size_t n, size_t batch,
                            size_t stride_coeffs);
int kzg_wait_batch(kzg_ctx* ctx, int slot, uint64_t* out_p1s /* batch x 18 */, size_t batch);
/* Batched Evaluation::generate_proof (reference src/polynomial.rs:260-269): polynomial p is opened at
 * zs[p] with claimed value ys[p] (4 x uint64 Montgomery each); one quotient scan per polynomial, one
 * batched MSM over the quotients.  n >= 2, n - 1 <= kzg_srs_len.  kzg_wait_open_batch fills one status per
 * polynomial (KZG_OK, KZG_ERR_CONSTANT_POLY, KZG_ERR_REMAINDER) and the proofs of those that are KZG_OK. */
int kzg_open_batch_submit(kzg_ctx* ctx, int slot, const void* d_coeffs, size_t n, size_t batch,
                          size_t stride_coeffs, const uint64_t* zs, const uint64_t* ys);
int kzg_wait_open_batch(kzg_ctx* ctx, int slot, uint64_t* out_p1s /* batch x 18 */, int* statuses, size_t batch);

/* raw device memory helpers so a non-HIP host (Rust, Python) can stage device-resident inputs */
int kzg_dev_alloc(kzg_ctx* ctx, size_t bytes, void** out_dptr);
int kzg_dev_free(kzg_ctx* ctx, void* dptr);
int kzg_dev_upload(kzg_ctx* ctx, void* dst_dptr, const void* src_host, size_t bytes);
int kzg_dev_download(kzg_ctx* ctx, void* dst_host, const void* src_dptr, size_t bytes);

/* ---- G1 helpers on the host side of the boundary ------------------------------------------ */

/* Sum of k blst_p1 values, normalised like kzg_commit's output.  This is the "reduce" of the
 * multi-GPU MSM: each rank commits its SRS slice, the partial sums are all-gathered over RCCL and
 * every rank (or rank 0) calls this (blst_p1_add_or_double semantics, reference src/curves.rs:79-85). */
int kzg_g1_sum(const uint64_t* p1s, size_t k, uint64_t out_p1[18]);

/* ZCash 48-byte compression = what the reference's `Serialize for G1Point` emits
 * (reference src/curves.rs:99-110 -> blst_p1_compress).  The parity comparator. */
int kzg_g1_compress(const uint64_t p1[18], uint8_t out[48]);

/* Inverse: what the reference's `Deserialize for G1Point` obtains from blst_p1_uncompress +
 * blst_p1_from_affine (reference src/curves.rs:112-183), e.g. to ingest the CLI's setup.json.  On-curve
 * check only, as blst.  Next-row component. */
int kzg_g1_uncompress(const uint8_t in[48], uint64_t out_p1[18]);

/* The G2 half of SetupArtifact `index` on the host: [s^index mod r]G2 as a blst_p2 (36 x u64: x, y, z in Fp2,
 * Montgomery, z = 1), s = secret read big-endian (reference src/trusted_setup.rs:20-28, 40-53, 64-72).  A verifier
 * only ever reads index 1 (src/polynomial.rs:284): this closes the commit -> open -> verify round trip of
 * src/lib.rs:16-33 for a host that has no blst.  ~0.5 ms. */
int kzg_srs_g2_at(const uint8_t secret_be[32], uint64_t index, uint64_t out_p2[36]);

/* Evaluation::verify_proof (reference src/polynomial.rs:276-294):
 *     e(proof, [s]G2 - [z]G2) == e(commitment - [y]G1, G2)
 * commitment, proof: blst_p1; z = evaluation.point, y = evaluation.result: blst_fr (Montgomery);
 * s_g2 = setup_artifacts[1].g2: blst_p2 (36 x u64: Jacobian x, y, z in Fp2, Montgomery).  Host only (both
 * pairings in one Miller loop, ~3 ms); *valid = 1 accepted, 0 rejected.  KZG_ERR_INVALID_ARG
 * when s_g2 is not on the curve.  Next-row component. */
int kzg_verify_proof(const uint64_t commitment_p1[18], const uint64_t proof_p1[18], const uint64_t z[4],
                     const uint64_t y[4], const uint64_t s_g2_p2[36], int* valid);
/* n independent checks against the same setup (BASELINE config 5: a batch of openings and their verification),
 * spread over the host's cores; element i of every array belongs to check i (18 / 18 / 4 / 4 u64, one int). */
int kzg_verify_proof_batch(const uint64_t* commitments_p1, const uint64_t* proofs_p1, const uint64_t* zs,
                           const uint64_t* ys, const uint64_t s_g2_p2[36], size_t n, int* valid);

/* ---- measurement -------------------------------------------------------------------------- */

typedef struct kzg_kernel_times {
    /* the most recent job collected from the slot.  accumulate_ms and references come from the device itself and are
     * filled for EVERY job; the other fields are HIP-event spans on the streams the kernels ran on, in milliseconds,
     * filled only for jobs submitted while timing was enabled (kzg_set_timing): a timed job records six events */
    float digits_ms;      /* scalar recoding + two-level counting sort (all of msm_sort.hip) */
    float scatter_ms;     /* queueing: buffer clears + the wait for the shared accumulation stream (not kernel cost) */
    float accumulate_ms;  /* bucket accumulation, the dominant kernel: its own duration, first wave in to last wave out on the
                           * device's constant 100 MHz clock (the kernel stamps itself; no event sits between two launches) */
    float reduce_ms;      /* finalisation + reduction trees, INCLUDING their wait behind the next accumulation */
    float quotient_ms;    /* open only: scalar-field synthetic division */
    float total_ms;       /* first kernel start -> last kernel end */
    uint64_t references;  /* non-zero scalar digits = mixed additions of the accumulation kernel (whole batch) */
    float accumulate_events_ms;  /* the HIP-event bracket around the same launch on its stream: kernel + the time it waited for the chip */
    float reserved_;
} kzg_kernel_times;

int kzg_set_timing(kzg_ctx* ctx, int enabled);
int kzg_get_times(kzg_ctx* ctx, int slot, kzg_kernel_times* out);
/* Scalar recoding chosen for the loaded SRS: recoding 0 = aligned signed windows of digit_bits
 * (table_levels = number of windows), 1 = width-digit_bits non-adjacent form over a table with one level per
 * scalar bit (table_levels = 255; chosen when that table fits the free HBM).  Any out pointer may be NULL. */
int kzg_msm_config(const kzg_ctx* ctx, int* digit_bits, int* table_levels, size_t* num_buckets, int* recoding);

#ifdef __cplusplus
}
#endif
#endif /* KZG_MI355X_H */
