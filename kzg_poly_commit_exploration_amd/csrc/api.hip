// api.hip -- the C-ABI of libkzg_mi355x.so (declared in include/kzg_mi355x.h): context, SRS
// residency, stream slots, the host-side orchestration of one commitment / opening, and the serial
// tail (a few dozen point additions + one inversion) that finishes each MSM on the host.
//
// One commitment on a slot's stream:
//   digits+histogram -> scan -> scatter -> bucket accumulation -> bucket finalisation
//   -> row / column tree sums of the bucket matrix, split once more (small jobs: one launch behind the sort)
//   -> D2H of <= 128 XYZZ partials -> host: four short weighted sums, normalise, blst_p1 out.
// Nothing here falls back to the CPU for the MSM or the division: without a device the context
// cannot be created.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "engine.h"
#include "host_field.hpp"
#include "host_fr.hpp"
#include "host_pairing.hpp"

using namespace kzg;
namespace hf = kzg_host;

namespace {

// The reference refuses polynomials of more than u32::MAX coefficients (src/polynomial.rs:56-61); the kernels index
// with 32 bits, so the C-ABI refuses them too instead of truncating the count.
constexpr size_t kMaxCoefficients = 0xFFFFFFFFull;

// Stream slots per context (+ the shared accumulation stream: 6 of the 8 hardware queues the library asks for).  The
// light kernels of a job -- sort in front of its accumulation, finalisation and reduction trees behind it -- only get
// the chip in the tail of ANOTHER job's accumulation and crawl while one is running (round-3 timeline: a 54 us sort
// kernel takes 260, the trees 1 ms).  With three jobs in flight the next job's sort regularly finished 100-240 us after
// the accumulation it should have hidden under; five give it one more accumulation's worth of time: 366 -> 389
// commitments/s on the same box (4: 383-390, 6: 381-385, 7: 368-385).
#ifndef KZG_NUM_SLOTS
#define KZG_NUM_SLOTS 5
#endif
constexpr int kNumSlots = KZG_NUM_SLOTS;
// Reduction plan (msm_reduce.hip): bucket index b = hi * C + lo; Row (R entries) and Col (C entries)
// are each split once more into a "row" part and a "column" part that the host receives.
struct ReducePlan {
    uint32_t lo_bits = 0, hi_bits = 0;  // C = 2^lo_bits, R = 2^hi_bits
    uint32_t row_lo = 0, row_hi = 0;    // split of the Row vector index (hi_bits = row_lo + row_hi)
    uint32_t col_lo = 0, col_hi = 0;    // split of the Col vector index (lo_bits = col_lo + col_hi)
    // record offsets inside the final buffer
    uint32_t off_r2row = 0, off_c2row = 0, off_r2col = 0, off_c2col = 0, total = 0;
};

// what a slot holds decides which wait entry point may collect it (a batched job's final buffer is laid out
// [section][polynomial][record]; reading it as a single job would return a wrong point with KZG_OK)
// SLOT_RESERVED: owned by a synchronous host-pointer call between its steps (upload -> submit -> wait), during which the
// context mutex is NOT held: N caller threads occupy N slots and their jobs pipeline like explicit submits do.
enum SlotKind { SLOT_IDLE = 0, SLOT_COMMIT = 1, SLOT_OPEN = 2, SLOT_TRIVIAL = 3, SLOT_COMMIT_BATCH = 4, SLOT_OPEN_BATCH = 5, SLOT_RESERVED = 6 };

struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    hipEvent_t done = nullptr;
    hipEvent_t sorted_ev = nullptr, accum_ev = nullptr;  // hand-offs to / from the shared accumulation stream
    // MSM workspace (sized at SRS load)
    uint32_t *d_cnt = nullptr, *d_offs = nullptr, *d_block_sums = nullptr;
    uint64_t* d_pairs = nullptr;
    uint32_t* d_sorted = nullptr;
    void* d_buckets = nullptr;
    void *d_part_a = nullptr, *d_part_b = nullptr;  // head / tail partials of the accumulation segments
    void* d_pair_scratch = nullptr;  // prefix products of the affine front end (msm_accum.hip)
    void* d_heavy_ws = nullptr;   // long-bucket registry + tree buffers of msm_finalize.hip
    void* d_arena = nullptr;     // Row[] then Col[] vectors of the bucket matrix
    // The <= 128 partial sums the host finishes are written by the last reduction kernel STRAIGHT into pinned host
    // memory (d_final is the device's address of h_final): a copy back is a blit kernel on this runtime, and behind an
    // accumulation kernel that fills the chip it took 1.3 ms instead of 5 us -- the host collected the job that much
    // later, submitted the next one that much later, and its sort finished after the accumulation it should have
    // hidden under (100-240 us of idle chip per commitment; round-3 timeline).
    void* d_final = nullptr;
    uint64_t* h_final = nullptr;  // pinned, mapped
    // polynomial workspace (grown on demand)
    size_t poly_cap = 0;
    uint32_t* d_stage = nullptr;  // coefficients copied from the host
    uint32_t* d_q = nullptr;      // quotient
    uint32_t* d_chunk = nullptr;
    uint32_t* d_block = nullptr;
    // The job's flag words live in pinned host memory that the kernels write directly (d_small is the device's address of
    // h_small; plain stores only -- every writer of a flag stores the same 1): no memset and no copy-back on the stream.
    // A 256-byte copy back is a blit kernel here, and queued behind a chip-filling accumulation it sat for ~1 ms between
    // the job's last kernel and the moment the host could collect it.
    uint32_t* d_small = nullptr;  // [0..1] flags, [8..15] P(z), [16..23] c0, [24] tail flag, [26] references
    uint32_t* h_small = nullptr;  // pinned, mapped
    uint32_t* d_bsmall = nullptr;  // batched openings: 32 words per polynomial, same layout as d_small[0..31]
    uint32_t* h_bsmall = nullptr;
    size_t bsmall_cap = 0;
    std::vector<uint32_t> open_ys;  // y of every polynomial of a batched opening (8 words each)
    // state of the job in flight
    SlotKind kind = SLOT_IDLE;
    size_t job_n = 0;
    uint32_t job_batch = 1;
    uint32_t open_y[8] = {};
    bool timing = false;
    bool has_quotient = false;
    bool tail_checked = false;
    kzg_kernel_times times = {};
};

}  // namespace

struct kzg_ctx {
    // a multi-device context (kzg_ctx_create_multi) only carries this pointer: its calls are sharded over the
    // single-device contexts inside (multi.hip)
    kzg::MultiState* multi = nullptr;
    int device = 0;
    std::mutex mu;
    std::condition_variable slot_cv;  // a slot became idle / a synchronous call finished
    int sync_owned = 0;               // slots currently owned by synchronous host-pointer calls (reserve_slot)
    bool raw_partials = false;        // kid of a range-split multi-device context: results stay un-normalised (ctx_set_raw_partials)
    // KZG_HOST_TRACE=1: where a synchronous host-pointer call spends its wall time (printed when the context is destroyed)
    bool host_trace = false;
    std::atomic<uint64_t> trace_ns[5] = {};  // reserve (waiting for a slot), upload, submit, device wait, collect (host tail)
    std::atomic<uint64_t> trace_calls{0};
    std::string last_error;
    // SRS
    size_t n = 0;  // points
    MsmConfig cfg = {};
    void* d_table = nullptr;  // W * n affine points
    ReducePlan plan;
    size_t arena_records = 0, final_records = 0;  // per polynomial of a batch
    uint32_t max_batch = 1;                        // polynomials per submit the workspaces are sized for
    Slot slots[kNumSlots];
    // All bucket-accumulation kernels run on ONE stream, in submission order: each fills the chip on its
    // own, so letting two of them overlap only makes both slower (and their timings meaningless), while
    // the light sort / reduction kernels of the other slots run beside it on the slots' own streams.
    hipStream_t heavy_stream = nullptr;
    bool serialize_accum = true;   // KZG_SERIALIZE_ACCUM=0 lets accumulation kernels of different slots overlap
    // LDS reserved per accumulation workgroup (KZG_ACCUM_LDS_KB overrides): 41 KB would cap the kernel at three workgroups
    // per CU; its register count (176 reserved, msm_accum.hip) caps it at two, which is what its grid is launched for.
    // Two workgroups leave 78 KB of LDS and 160 VGPRs per SIMD lane to the light kernels of the other slots (32 KB instead
    // of 41 measured the same).
    uint32_t accum_lds_bytes = 41u * 1024u;
    uint32_t small_lds_bytes = 48u * 1024u;  // k_small_msm's LDS reservation (raised to small_msm_lds_bytes() at creation)
    bool small_msm_off = false;              // KZG_SMALL_MSM=0: small jobs take the general multi-launch path (A/B, tests)
    bool slots_ready = false;
    bool timing = false;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                  \
            return KZG_ERR_HIP;                                                                     \
        }                                                                                           \
    } while (0)

// scope guards for the setup paths (several HIP_TRY early returns)
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
};
struct TmpStream {
    hipStream_t s = nullptr;
    ~TmpStream() { if (s) hipStreamDestroy(s); }
};

void free_slot_msm(Slot& s) {
    hipFree(s.d_cnt); hipFree(s.d_offs); hipFree(s.d_block_sums); hipFree(s.d_pairs); hipFree(s.d_sorted);
    hipFree(s.d_buckets); hipFree(s.d_part_a); hipFree(s.d_part_b); hipFree(s.d_heavy_ws); hipFree(s.d_arena);
    hipFree(s.d_pair_scratch);
    s.d_pair_scratch = nullptr;
    if (s.h_final) hipHostFree(s.h_final);
    s.d_cnt = s.d_offs = s.d_block_sums = s.d_sorted = nullptr;
    s.d_pairs = nullptr;
    s.d_buckets = s.d_part_a = s.d_part_b = s.d_arena = s.d_final = nullptr;
    s.d_heavy_ws = nullptr;
    s.h_final = nullptr;
}
void free_slot_poly(Slot& s) {
    hipFree(s.d_stage); hipFree(s.d_q); hipFree(s.d_chunk); hipFree(s.d_block);
    s.d_stage = s.d_q = s.d_chunk = s.d_block = nullptr;
    s.poly_cap = 0;
}

// stream, events and the small flag buffers of a slot (what kzg_quotient / kzg_evaluate need without an SRS)
int ensure_slot_basics(kzg_ctx* ctx, Slot& s) {
    if (s.stream) return KZG_OK;
    {
        // slot streams run the light kernels (sort, finalise, reduction, quotient): lowest priority, so that
        // when an accumulation ends the NEXT accumulation (high-priority stream) takes the chip first and the
        // ~180-VGPR reduction kernels fill in behind it instead of holding half of every SIMD's registers
        int least = 0, greatest = 0;
        HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        const char* v = std::getenv("KZG_STREAM_PRIORITIES");
        const bool prio = !(v && v[0] == '0');
        HIP_TRY(ctx, hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, prio ? least : 0));
    }
    for (auto& e : s.ev) HIP_TRY(ctx, hipEventCreate(&e));
    HIP_TRY(ctx, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    HIP_TRY(ctx, hipHostMalloc(&s.h_small, 64 * 4, hipHostMallocMapped));
    HIP_TRY(ctx, hipHostGetDevicePointer((void**)&s.d_small, s.h_small, 0));
    return KZG_OK;
}

int ensure_poly(kzg_ctx* ctx, Slot& s, size_t n) {
    if (n <= s.poly_cap && s.d_stage) return KZG_OK;
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    free_slot_poly(s);
    size_t cap = n < 1024 ? 1024 : n;
    HIP_TRY(ctx, hipMalloc(&s.d_stage, cap * 32));
    HIP_TRY(ctx, hipMalloc(&s.d_q, cap * 32));
    HIP_TRY(ctx, hipMalloc(&s.d_chunk, poly_chunk_words((uint32_t)cap) * 4));
    HIP_TRY(ctx, hipMalloc(&s.d_block, poly_block_words((uint32_t)cap) * 4));
    s.poly_cap = cap;
    return KZG_OK;
}

// reduction plan: depends only on the bucket count
void plan_reduce(kzg_ctx* ctx) {
    ReducePlan P;
    uint32_t bits = 0;  // log2(buckets)
    while ((1u << bits) < ctx->cfg.nb) bits++;
    P.lo_bits = bits / 2;
    P.hi_bits = bits - P.lo_bits;
    P.row_lo = P.hi_bits / 2;
    P.row_hi = P.hi_bits - P.row_lo;
    P.col_lo = P.lo_bits / 2;
    P.col_hi = P.lo_bits - P.col_lo;
    P.off_r2row = 0;
    P.off_c2row = P.off_r2row + (1u << P.row_hi);
    P.off_r2col = P.off_c2row + (1u << P.row_lo);
    P.off_c2col = P.off_r2col + (1u << P.col_hi);
    P.total = P.off_c2col + (1u << P.col_lo);
    ctx->plan = P;
    ctx->arena_records = ((size_t)1 << P.hi_bits) + ((size_t)1 << P.lo_bits);
    ctx->final_records = P.total;
}

int setup_slots_impl(kzg_ctx* ctx);
// (Re)allocates every slot's MSM workspace.  On any failure the context is left WITHOUT an SRS (n = 0, no
// workspaces, slots_ready false): the next commit returns KZG_ERR_NO_SRS instead of launching on freed buffers.
int setup_slots(kzg_ctx* ctx, bool keep_table_on_failure = false) {
    ctx->slots_ready = false;
    int rc = setup_slots_impl(ctx);
    if (rc != KZG_OK) {
        for (auto& s : ctx->slots) free_slot_msm(s);
        if (keep_table_on_failure) return rc;  // the caller retries with the previous sizes
        if (ctx->d_table) {
            hipFree(ctx->d_table);
            ctx->d_table = nullptr;
        }
        ctx->n = 0;
        (void)hipGetLastError();
    }
    return rc;
}
int setup_slots_impl(kzg_ctx* ctx) {
    plan_reduce(ctx);
    if (const char* v = std::getenv("KZG_TEST_FAIL_SLOT_ALLOC")) {  // fault injection for tests/test_gpu_parity.py
        const int mode = std::atoi(v);  // 1: every setup fails; 2: only setups for more than one polynomial per batch
        if (mode == 1 || (mode == 2 && ctx->max_batch > 1)) {
            ctx->last_error = "injected allocation failure (KZG_TEST_FAIL_SLOT_ALLOC)";
            return KZG_ERR_HIP;
        }
    }
    if (!ctx->heavy_stream) {
        int least = 0, greatest = 0;
        HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        const char* v = std::getenv("KZG_STREAM_PRIORITIES");
        const bool prio = !(v && v[0] == '0');
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->heavy_stream, hipStreamNonBlocking, prio ? greatest : 0));
    }
    const MsmConfig cfg = ctx->cfg;
    if (ctx->max_batch > sort_max_batch(cfg)) ctx->max_batch = sort_max_batch(cfg);
    const size_t B = ctx->max_batch;
    const size_t pairs = (size_t)cfg.max_digits * ctx->n * B;
    for (int i = 0; i < kNumSlots; i++) {
        Slot& s = ctx->slots[i];
        {
            int rcb = ensure_slot_basics(ctx, s);
            if (rcb) return rcb;
        }
        if (!s.sorted_ev) {
            HIP_TRY(ctx, hipEventCreateWithFlags(&s.sorted_ev, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&s.accum_ev, hipEventDisableTiming));
        }
        free_slot_msm(s);
        HIP_TRY(ctx, hipMalloc(&s.d_cnt, (size_t)sort_count_entries((uint32_t)B, cfg) * 4 + 64));
        HIP_TRY(ctx, hipMalloc(&s.d_offs, ((size_t)cfg.nb * B + 1) * 4));
        HIP_TRY(ctx, hipMalloc(&s.d_block_sums, (size_t)sort_workspace_words() * 4));  // bin starts, chunk plan, fill counters, fine-pass table
        HIP_TRY(ctx, hipMemset(s.d_block_sums, 0, (size_t)sort_workspace_zero_words() * 4));
        HIP_TRY(ctx, hipMalloc(&s.d_pairs, (pairs ? pairs : 1) * 8));
        HIP_TRY(ctx, hipMalloc(&s.d_sorted, (pairs ? pairs : 1) * 4));
        HIP_TRY(ctx, hipMalloc(&s.d_buckets, (size_t)cfg.nb * B * kXyzzBytes));
        HIP_TRY(ctx, hipMalloc(&s.d_part_a, (size_t)kMaxAccumLanes * kXyzzBytes));
        HIP_TRY(ctx, hipMalloc(&s.d_part_b, (size_t)kMaxAccumLanes * kXyzzBytes));
        if (const size_t pair_bytes = accumulate_pair_scratch_bytes(pairs))  // only with KZG_ACCUM_PAIRS=1
            HIP_TRY(ctx, hipMalloc(&s.d_pair_scratch, pair_bytes));
        HIP_TRY(ctx, hipMalloc(&s.d_heavy_ws, heavy_workspace_bytes()));
        HIP_TRY(ctx, hipMalloc(&s.d_arena, ctx->arena_records * B * kXyzzBytes));
        HIP_TRY(ctx, hipHostMalloc(&s.h_final, ctx->final_records * B * kXyzzBytes, hipHostMallocMapped));
        HIP_TRY(ctx, hipHostGetDevicePointer(&s.d_final, s.h_final, 0));
        s.kind = SLOT_IDLE;
    }
    ctx->slots_ready = true;
    return KZG_OK;
}

// builds levels 1..W-1 of the table from level 0 (already in d_table[0..n))
int build_tables(kzg_ctx* ctx, hipStream_t st, void* d_xyzz_tmp, void* d_prefix) {
    const size_t n = ctx->n;
    for (uint32_t j = 1; j < ctx->cfg.W; j++) {
        char* prev = (char*)ctx->d_table + (size_t)(j - 1) * n * kAffineBytes;
        char* next = (char*)ctx->d_table + (size_t)j * n * kAffineBytes;
        launch_table_window(st, prev, (uint32_t)n, ctx->cfg.level_bits, d_xyzz_tmp, d_prefix, next);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return KZG_OK;
}

int drain_all(kzg_ctx* ctx) {
    if (ctx->heavy_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->heavy_stream));
    for (auto& s : ctx->slots)
        if (s.stream) HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    return KZG_OK;
}

int srs_prepare(kzg_ctx* ctx, size_t n) {
    if (n == 0 || n > 0x7fffffffu / 32) return KZG_ERR_INVALID_ARG;
    ctx->slots_ready = false;
    int rc = drain_all(ctx);
    if (rc) return rc;
    for (auto& s : ctx->slots) s.kind = SLOT_IDLE;
    if (ctx->d_table) {
        hipFree(ctx->d_table);
        ctx->d_table = nullptr;
    }
    ctx->n = 0;
    // The opt-in NAF recoding wants a 255-level table (engine.h); it may take up to 60 % of the HBM that is free
    // now (KZG_TABLE_GB overrides), otherwise the windowed recoding with its ~15 levels is used.
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    size_t budget = (size_t)((double)free_b * 0.6);
    if (const char* v = std::getenv("KZG_TABLE_GB")) budget = (size_t)(std::strtod(v, nullptr) * 1073741824.0);
    MsmConfig cfg = choose_msm_config(n, budget);
    if ((size_t)cfg.W * n >= 0x80000000ull) return KZG_ERR_INVALID_ARG;  // table index must fit 31 bits
    ctx->cfg = cfg;
    HIP_TRY(ctx, hipMalloc(&ctx->d_table, (size_t)cfg.W * n * kAffineBytes));
    return KZG_OK;
}

// enqueue `batch` MSMs of n scalars each (polynomial p at d_scalars + p * stride scalars) on slot s
int enqueue_msm(kzg_ctx* ctx, Slot& s, const uint32_t* d_scalars, int is_mont, size_t n, int ev_base,
                uint32_t batch = 1, uint64_t stride = 0) {
    const MsmConfig cfg = ctx->cfg;
    const uint32_t nbt = cfg.nb * batch;  // polynomial-major bucket ids
    hipStream_t st = s.stream;
    if (s.timing) HIP_TRY(ctx, hipEventRecord(s.ev[ev_base], st));
    const bool header_zeroed = launch_bucket_sort(st, d_scalars, is_mont, (uint32_t)n, batch, stride, (uint32_t)ctx->n, cfg, s.d_cnt,
                       s.d_block_sums, s.d_pairs, s.d_offs, s.d_sorted, (uint32_t*)s.d_heavy_ws);
    if (s.timing) {
        HIP_TRY(ctx, hipEventRecord(s.ev[ev_base + 1], st));
        HIP_TRY(ctx, hipEventRecord(s.ev[ev_base + 2], st));
    }
    bool alone = true;  // (this slot is still marked idle while its job is being enqueued)
    for (const auto& other : ctx->slots) alone = alone && (&other == &s || other.kind == SLOT_IDLE);
    const uint64_t max_refs = (uint64_t)n * cfg.max_digits * batch;
    const uint32_t lanes = accumulate_lanes(max_refs, alone);
    // reduction: Row / Col tree sums of every polynomial's bucket matrix, each split once more.
    // Vectors are polynomial-major ([p][index]); the final buffer holds four sections [p][len_k].
    const ReducePlan& P = ctx->plan;
    const uint32_t R = 1u << P.hi_bits, C = 1u << P.lo_bits, B = batch;
    char* row = (char*)s.d_arena;
    char* col = row + (size_t)R * B * kXyzzBytes;
    char* fin = (char*)s.d_final;
    const uint32_t rl = 1u << P.row_lo, rh = 1u << P.row_hi, cl = 1u << P.col_lo, ch = 1u << P.col_hi;
    const TreeSumDesc stage1[2] = {
        {s.d_buckets, row, B * R, C, C, 1, B * R, 0},           // Row[p][hi] = sum_lo Bk[p][hi*C + lo]
        {s.d_buckets, col, B * C, R, 1, C, C, (uint64_t)cfg.nb}};  // Col[p][lo] = sum_hi Bk[p][hi*C + lo]
    const TreeSumDesc stage2[4] = {
        {row, fin + (size_t)P.off_r2row * B * kXyzzBytes, B * rh, rl, rl, 1, rh, R},
        {row, fin + (size_t)P.off_c2row * B * kXyzzBytes, B * rl, rh, 1, rl, rl, R},
        {col, fin + (size_t)P.off_r2col * B * kXyzzBytes, B * ch, cl, cl, 1, ch, C},
        {col, fin + (size_t)P.off_c2col * B * kXyzzBytes, B * cl, ch, 1, cl, cl, C}};
    if (!header_zeroed) HIP_TRY(ctx, hipMemsetAsync(s.d_heavy_ws, 0, kHeavyHeaderBytes, st));  // long-bucket counters, phase counters
    if (max_refs <= kTinyRefs && !ctx->small_msm_off) {
        // Small jobs are chains of dependent additions on a nearly empty chip: everything from here to the copy back
        // in ONE launch on the slot's own stream (msm_finalize.hip: k_small_msm), no bucket memset, no stream hand-over.
        if (s.timing) HIP_TRY(ctx, hipEventRecord(s.ev[ev_base + 3], st));
        launch_small_msm(st, ctx->d_table, s.d_sorted, s.d_offs, nbt, lanes, max_refs, s.d_buckets, s.d_part_a, s.d_part_b,
                         s.d_heavy_ws, s.d_small + 26, stage1, stage2, ctx->small_lds_bytes);
        if (s.timing) HIP_TRY(ctx, hipEventRecord(s.ev[ev_base + 4], st));
    } else {
        // hand over to the shared accumulation stream and back (each hand-over costs ~12 us: not when no other slot
        // has work whose accumulation this one could collide with)
        const bool hand_over = ctx->serialize_accum && !alone;
        hipStream_t hs = hand_over ? ctx->heavy_stream : st;
        if (hand_over) {
            HIP_TRY(ctx, hipEventRecord(s.sorted_ev, st));
            HIP_TRY(ctx, hipStreamWaitEvent(hs, s.sorted_ev, 0));
        }
        if (s.timing) HIP_TRY(ctx, hipEventRecord(s.ev[ev_base + 3], hs));
        launch_bucket_accumulate(hs, ctx->d_table, s.d_sorted, s.d_offs, nbt, lanes, s.d_buckets, s.d_part_a, s.d_part_b,
                                 ctx->accum_lds_bytes, s.d_pair_scratch, max_refs,
                                 (char*)s.d_heavy_ws + kAccumClockOffset);
        if (s.timing) HIP_TRY(ctx, hipEventRecord(s.ev[ev_base + 4], hs));
        if (hand_over) {
            HIP_TRY(ctx, hipEventRecord(s.accum_ev, hs));
            HIP_TRY(ctx, hipStreamWaitEvent(st, s.accum_ev, 0));
        }
        launch_bucket_finalize(st, s.d_offs, nbt, lanes, s.d_part_a, s.d_part_b, s.d_buckets, s.d_heavy_ws, s.d_small + 26,
                               finalize_group_size(nbt));
        // (Rounds 1-2 put a gate kernel here that deferred the reduction trees to the tail of the next slot's
        // accumulation: worth 20 % with round 1's 206-VGPR accumulation kernel, beside which the trees could become
        // resident; the trees (243-258 VGPRs) cannot start beside round 3's accumulation kernel whatever the stream
        // order says, and with or without a gate -- LDS-sized or a device-side count of running workgroups -- round 3
        // measured 409 / 384 against 408 / 383 and 414 / 386 against 415 / 386 commitments / proofs per second.  Retired.)
        launch_tree_sums_two_stage(st, stage1, 2, stage2, 4, (uint32_t*)s.d_heavy_ws + 64, alone);
    }
    if (s.timing) HIP_TRY(ctx, hipEventRecord(s.ev[ev_base + 5], st));
    HIP_TRY(ctx, hipGetLastError());
    return KZG_OK;
}

// host tail.  With V indexed by u = u1 * 2^lo + u0:  sum_u u V_u = 2^lo * wsum(R2) + wsum(C2), where
// R2[u1] = sum_u0 V, C2[u0] = sum_u1 V and wsum(P) = sum_v v * P_v (running sums, <= 32 entries).
hf::PX host_wsum(const uint64_t* recs, uint32_t len) {
    hf::PX run = hf::px_inf(), acc = hf::px_inf();
    for (uint32_t v = len; v-- > 1;) {
        run = hf::px_add(run, hf::px_from_record(recs + (size_t)v * kXyzzWords64));
        acc = hf::px_add(acc, run);
    }
    return acc;
}
hf::PX host_shift(hf::PX p, uint32_t k) {
    for (uint32_t i = 0; i < k; i++) p = hf::px_double(p);
    return p;
}
hf::P1 finish_msm(const kzg_ctx* ctx, const Slot& s, uint32_t p = 0, uint32_t batch = 1) {
    const ReducePlan& P = ctx->plan;
    const uint64_t* f = s.h_final;
    // section k of the final buffer is [batch][len_k]: polynomial p's block starts at off_k*batch + p*len_k
    auto at = [&](uint32_t off, uint32_t len) { return f + ((size_t)off * batch + (size_t)p * len) * kXyzzWords64; };
    const uint32_t rh = 1u << P.row_hi, rl = 1u << P.row_lo, ch = 1u << P.col_hi, cl = 1u << P.col_lo;
    // W(Row) = 2^row_lo * wsum(R2row) + wsum(C2row);  W(Col) likewise
    hf::PX w_row = hf::px_add(host_shift(host_wsum(at(P.off_r2row, rh), rh), P.row_lo), host_wsum(at(P.off_c2row, rl), rl));
    hf::PX w_col = hf::px_add(host_shift(host_wsum(at(P.off_r2col, ch), ch), P.col_lo), host_wsum(at(P.off_c2col, cl), cl));
    // sum_b b B_b = C * W(Row) + W(Col);  sum_b B_b = sum of R2row.
    // Bucket b weighs b + 1 (windows: digit magnitude) or 2b + 1 (NAF: odd digits only).
    hf::PX total = hf::px_add(host_shift(w_row, P.lo_bits), w_col);
    if (ctx->cfg.recode == kRecodeNaf) total = hf::px_double(total);
    const uint64_t* r2 = at(P.off_r2row, rh);
    for (uint32_t k = 0; k < rh; k++) total = hf::px_add(total, hf::px_from_record(r2 + (size_t)k * kXyzzWords64));
    // a kid of a range-split multi-device context hands its partial sum on as it is (Jacobian, two products); the
    // parent pays the one inversion after adding the K partials (multi.hip)
    return ctx->raw_partials ? hf::px_to_jacobian(total) : hf::px_normalize(total);
}

void write_p1(uint64_t out[18], const hf::P1& p) { std::memcpy(out, &p, sizeof p); }

// EVERY job: accumulate_ms is the accumulation kernel's own duration -- first wave in to last wave out on the constant
// 100 MHz clock, stamped by the kernel and passed on by the finalisation (h_small[28..31]; the one-launch path of
// small jobs leaves it 0) -- and `references` its number of mixed additions: no stream event is involved.  Timed jobs
// (kzg_set_timing) also get the HIP-event spans; accumulate_events_ms is the bracket around the same launch on its
// stream, which holds the time the launch waited for the chip as well.
void fill_device_times(Slot& s) {
    s.times.references = s.h_small[26];
    const uint64_t not_start = (uint64_t)s.h_small[28] | ((uint64_t)s.h_small[29] << 32);
    const uint64_t end = (uint64_t)s.h_small[30] | ((uint64_t)s.h_small[31] << 32);
    if (not_start != 0 && end > ~not_start) s.times.accumulate_ms = (float)((double)(end - ~not_start) * 1e-5);  // 10 ns ticks
}
void fill_accumulate_times(Slot& s) {  // timed jobs, after fill_device_times
    float ms = 0;
    (void)hipEventElapsedTime(&ms, s.ev[3], s.ev[4]);
    s.times.accumulate_events_ms = ms;
    if (s.times.accumulate_ms == 0) s.times.accumulate_ms = ms;
}

// ---- slot ownership of the synchronous host-pointer entry points (all of these with ctx->mu held) ------------------
void slot_idle(kzg_ctx* ctx, Slot& s) {
    s.kind = SLOT_IDLE;
    ctx->slot_cv.notify_all();
}
// an idle slot, marked SLOT_RESERVED for the caller; -1 when none is free and (block == false, or every busy slot
// belongs to an explicit submit whose owner may be this very thread: waiting for it could never end -> KZG_ERR_BUSY)
int reserve_slot(kzg_ctx* ctx, std::unique_lock<std::mutex>& lk, bool block) {
    for (;;) {
        for (int i = 0; i < kNumSlots; i++)
            if (ctx->slots[i].kind == SLOT_IDLE) {
                ctx->slots[i].kind = SLOT_RESERVED;
                ctx->sync_owned++;
                return i;
            }
        if (!block || ctx->sync_owned == 0) return -1;
        ctx->slot_cv.wait(lk);
    }
}
void release_owned(kzg_ctx* ctx, int slot) {
    if (ctx->slots[slot].kind == SLOT_RESERVED) ctx->slots[slot].kind = SLOT_IDLE;
    ctx->sync_owned--;
    ctx->slot_cv.notify_all();
}
struct SlotLease {  // declared after the lock it lives under: released first
    kzg_ctx* ctx;
    int slot;
    ~SlotLease() { if (slot >= 0) release_owned(ctx, slot); }
};
// SRS replacement and workspace resizing wait until no synchronous call is between its steps
void quiesce(kzg_ctx* ctx, std::unique_lock<std::mutex>& lk) {
    while (ctx->sync_owned > 0) ctx->slot_cv.wait(lk);
}

bool host_tail_nonzero(const uint64_t* coeffs, size_t from, size_t n) {
    for (size_t i = from; i < n; i++)
        if (coeffs[4 * i] | coeffs[4 * i + 1] | coeffs[4 * i + 2] | coeffs[4 * i + 3]) return true;
    return false;
}

}  // namespace

// a tiny kernel for the device-pointer entry points: any non-zero Fr in [from, n)?
__global__ void k_tail_nonzero(const uint32_t* __restrict__ c, uint64_t from, uint64_t n, uint32_t* __restrict__ flag) {
    uint64_t i = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* p = reinterpret_cast<const uint4*>(c) + 2 * i;
    uint4 a = p[0], b = p[1];
    if (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) *flag = 1u;  // (host-mapped word: plain store, every writer stores 1)
}

// HIP multiplexes the streams of a process onto 4 hardware queues by default.  A context uses 4 streams of its own
// (3 slots + the accumulation stream); as soon as the host application (or RCCL, or torch) has streams too, two of
// them share a queue and a slot's reduction serialises in front of the next accumulation kernel (1.2 ms bubbles,
// tools/gaps.py).  The variable is read when the HIP runtime initialises, so it is set when this library is loaded
// (never overriding the user's own setting); a host that initialises HIP before loading the library has to export
// GPU_MAX_HW_QUEUES=8 itself.
__attribute__((constructor)) static void kzg_library_loaded() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

// asynchronous / device-pointer entry points belong to ONE device: refused on a multi-device context
#define KZG_SINGLE_DEVICE_ONLY(ctx)                                                                        \
    do {                                                                                                     \
        if ((ctx) && (ctx)->multi) {                                                                         \
            (ctx)->last_error = "this entry point takes a single-device context (kzg_ctx_create)";           \
            return KZG_ERR_INVALID_ARG;                                                                      \
        }                                                                                                    \
    } while (0)

extern "C" {

const char* kzg_strerror(int status) {
    switch (status) {
        case KZG_OK: return "ok";
        case KZG_ERR_DEGREE_TOO_HIGH:
            return "Setup does not allow for commitment generation of the polynomial. The polynomial degree is too high.";
        case KZG_ERR_CONSTANT_POLY: return "Unable to divide a constant polynomial";
        case KZG_ERR_REMAINDER:
            return "[divide_by_root] Fail to divide the polynomial by a root, constant terms do not add up";
        case KZG_ERR_INVALID_ARG: return "invalid argument";
        case KZG_ERR_NO_DEVICE: return "no HIP device available (this library has no CPU path)";
        case KZG_ERR_HIP: return "HIP runtime error";
        case KZG_ERR_NO_SRS: return "no SRS loaded";
        case KZG_ERR_BUSY: return "slot busy";
        default: return "unknown status";
    }
}

const char* kzg_last_error(const kzg_ctx* ctx) {
    if (!ctx) return "";
    if (ctx->multi && ctx->last_error.empty()) return multi_last_error(ctx->multi);
    return ctx->last_error.c_str();
}

int kzg_ctx_create_multi_ex(const int* devices, int ndev, unsigned flags, kzg_ctx** out) {
    if (!out) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    if (!devices || ndev <= 0 || (flags & ~(unsigned)KZG_MULTI_REPLICATE_SRS)) return KZG_ERR_INVALID_ARG;
    kzg::MultiState* m = nullptr;
    std::string err;
    int rc = multi_create(devices, ndev, (flags & KZG_MULTI_REPLICATE_SRS) ? kMultiReplicate : kMultiRange, &m, err);
    if (rc != KZG_OK) return rc;
    kzg_ctx* ctx = new kzg_ctx();
    ctx->multi = m;
    ctx->device = devices[0];
    *out = ctx;
    return KZG_OK;
}
int kzg_ctx_create_multi(const int* devices, int ndev, kzg_ctx** out) { return kzg_ctx_create_multi_ex(devices, ndev, 0u, out); }
int kzg_abi_version(void) { return KZG_ABI_VERSION; }
int kzg_num_devices(const kzg_ctx* ctx) { return !ctx ? 0 : (ctx->multi ? multi_num_devices(ctx->multi) : 1); }
uint64_t kzg_rccl_exchanges(const kzg_ctx* ctx) { return (ctx && ctx->multi) ? multi_rccl_exchanges(ctx->multi) : 0; }

int kzg_ctx_create(int device, kzg_ctx** out) {
    if (!out) return KZG_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return KZG_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return KZG_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return KZG_ERR_NO_DEVICE;
    kzg_ctx* ctx = new kzg_ctx();
    ctx->device = device;
    if (const char* v = std::getenv("KZG_SERIALIZE_ACCUM")) ctx->serialize_accum = std::atoi(v) != 0;
    if (const char* v = std::getenv("KZG_ACCUM_LDS_KB")) ctx->accum_lds_bytes = (uint32_t)std::atoi(v) * 1024u;
    if (const char* v = std::getenv("KZG_SMALL_MSM")) ctx->small_msm_off = std::atoi(v) == 0;
    if (const char* v = std::getenv("KZG_HOST_TRACE")) ctx->host_trace = std::atoi(v) != 0;
    if (hipFuncSetAttribute(small_msm_kernel(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_msm_lds_bytes()) == hipSuccess)
        ctx->small_lds_bytes = small_msm_lds_bytes();
    else
        (void)hipGetLastError();  // stays at 48 KiB: two workgroups may then share a CU (slower, not wrong)
    if (!poly_prepare_device()) {
        delete ctx;
        return KZG_ERR_HIP;  // the quotient kernels could not be launched on this device
    }
    *out = ctx;
    return KZG_OK;
}

void kzg_ctx_destroy(kzg_ctx* ctx) {
    if (!ctx) return;
    if (ctx->multi) {
        multi_destroy(ctx->multi);
        delete ctx;
        return;
    }
    hipSetDevice(ctx->device);
    if (ctx->host_trace && ctx->trace_calls.load()) {
        const double k = 1e-3 / (double)ctx->trace_calls.load();
        std::fprintf(stderr, "[kzg host trace] %llu kzg_commit calls, us per call: wait-for-slot %.1f upload %.1f submit %.1f device %.1f collect %.1f\n",
                     (unsigned long long)ctx->trace_calls.load(), k * ctx->trace_ns[0].load(), k * ctx->trace_ns[1].load(),
                     k * ctx->trace_ns[2].load(), k * ctx->trace_ns[3].load(), k * ctx->trace_ns[4].load());
    }
    for (auto& s : ctx->slots) {
        if (s.stream) hipStreamSynchronize(s.stream);
        free_slot_msm(s);
        free_slot_poly(s);
        if (s.h_small) hipHostFree(s.h_small);
        if (s.h_bsmall) hipHostFree(s.h_bsmall);
        for (auto& e : s.ev)
            if (e) hipEventDestroy(e);
        if (s.done) hipEventDestroy(s.done);
        if (s.sorted_ev) hipEventDestroy(s.sorted_ev);
        if (s.accum_ev) hipEventDestroy(s.accum_ev);
        if (s.stream) hipStreamDestroy(s.stream);
    }
    if (ctx->heavy_stream) hipStreamDestroy(ctx->heavy_stream);
    if (ctx->d_table) hipFree(ctx->d_table);
    delete ctx;
}

size_t kzg_srs_len(const kzg_ctx* ctx) { return !ctx ? 0 : (ctx->multi ? multi_srs_len(ctx->multi) : ctx->n); }
int kzg_num_slots(const kzg_ctx*) { return kNumSlots; }

int kzg_msm_config(const kzg_ctx* ctx, int* digit_bits, int* table_levels, size_t* num_buckets, int* recoding) {
    if (ctx && ctx->multi)  // the first slice's configuration (all slices have the same length up to one point)
        return kzg_msm_config(multi_kid(ctx->multi, 0), digit_bits, table_levels, num_buckets, recoding);
    if (!ctx || !ctx->n) return KZG_ERR_NO_SRS;
    if (digit_bits) *digit_bits = (int)ctx->cfg.c;
    if (table_levels) *table_levels = (int)ctx->cfg.W;
    if (num_buckets) *num_buckets = ctx->cfg.nb;
    if (recoding) *recoding = (int)ctx->cfg.recode;
    return KZG_OK;
}

int kzg_srs_load_g1(kzg_ctx* ctx, const void* first_g1, size_t stride, size_t n) {
    if (!ctx || !first_g1 || stride < 144) return KZG_ERR_INVALID_ARG;
    if (ctx->multi) {
        std::lock_guard<std::mutex> lkm(ctx->mu);
        ctx->last_error.clear();
        return n ? multi_srs_load(ctx->multi, first_g1, stride, n) : KZG_ERR_INVALID_ARG;
    }
    std::unique_lock<std::mutex> lk(ctx->mu);
    quiesce(ctx, lk);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = srs_prepare(ctx, n);
    if (rc) return rc;
    // gather the strided blst_p1 values, upload, normalise to affine = window 0
    std::vector<uint64_t> packed(n * 18);
    for (size_t i = 0; i < n; i++) std::memcpy(&packed[i * 18], (const char*)first_g1 + i * stride, 144);
    DevBuf d_jac, d_prefix, d_xyzz;
    HIP_TRY(ctx, hipMalloc(&d_jac.p, n * 144));
    HIP_TRY(ctx, hipMalloc(&d_prefix.p, n * 64));
    HIP_TRY(ctx, hipMalloc(&d_xyzz.p, n * kXyzzBytes));
    TmpStream st;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
    HIP_TRY(ctx, hipMemcpyAsync(d_jac.p, packed.data(), n * 144, hipMemcpyHostToDevice, st.s));
    ctx->n = n;
    launch_jacobian_to_affine(st.s, d_jac.p, (uint32_t)n, ctx->d_table, d_prefix.p);
    rc = build_tables(ctx, st.s, d_xyzz.p, d_prefix.p);
    if (rc) {
        ctx->n = 0;
        return rc;
    }
    return setup_slots(ctx);
}

int kzg_srs_generate_g1(kzg_ctx* ctx, const uint8_t secret_be[32], uint64_t first, size_t n) {
    if (!ctx || !secret_be) return KZG_ERR_INVALID_ARG;
    if (ctx->multi) {
        std::lock_guard<std::mutex> lkm(ctx->mu);
        ctx->last_error.clear();
        return n ? multi_srs_generate(ctx->multi, secret_be, first, n) : KZG_ERR_INVALID_ARG;
    }
    std::unique_lock<std::mutex> lk(ctx->mu);
    quiesce(ctx, lk);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = srs_prepare(ctx, n);
    if (rc) return rc;
    uint32_t raw[8];
    for (int w = 0; w < 8; w++) {
        // big-endian bytes -> little-endian 32-bit words (reference src/trusted_setup.rs:24)
        const uint8_t* b = secret_be + 28 - 4 * w;
        raw[w] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3];
    }
    size_t tmp_records = n > 32 * 255 ? n : 32 * 255;
    DevBuf d_gtable, d_prefix, d_xyzz;
    HIP_TRY(ctx, hipMalloc(&d_gtable.p, srs_gtable_bytes()));
    HIP_TRY(ctx, hipMalloc(&d_prefix.p, tmp_records * 64));
    HIP_TRY(ctx, hipMalloc(&d_xyzz.p, tmp_records * kXyzzBytes));
    TmpStream st;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
    ctx->n = n;
    launch_srs_generate(st.s, raw, first, (uint32_t)n, d_gtable.p, d_xyzz.p, d_prefix.p, ctx->d_table);
    rc = build_tables(ctx, st.s, d_xyzz.p, d_prefix.p);
    if (rc) {
        ctx->n = 0;
        return rc;
    }
    return setup_slots(ctx);
}

// level 0 is in d_table (builder's form): build the other levels, convert, size the slots
static int finish_srs_from_level0(kzg_ctx* ctx, hipStream_t st, size_t n) {
    DevBuf d_prefix, d_xyzz;
    HIP_TRY(ctx, hipMalloc(&d_prefix.p, n * 64));
    HIP_TRY(ctx, hipMalloc(&d_xyzz.p, n * kXyzzBytes));
    ctx->n = n;
    int rc = build_tables(ctx, st, d_xyzz.p, d_prefix.p);
    if (rc) {
        ctx->n = 0;
        return rc;
    }
    return setup_slots(ctx);
}

int kzg_srs_load_affine(kzg_ctx* ctx, const void* affine_xy, size_t n) {
    if (!ctx || !affine_xy || !n) return KZG_ERR_INVALID_ARG;
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (ctx->multi) {
        ctx->last_error.clear();
        return multi_srs_load_affine(ctx->multi, affine_xy, n);
    }
    quiesce(ctx, lk);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = srs_prepare(ctx, n);
    if (rc) return rc;
    DevBuf d_in;
    HIP_TRY(ctx, hipMalloc(&d_in.p, n * 96));
    TmpStream st;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
    HIP_TRY(ctx, hipMemcpyAsync(d_in.p, affine_xy, n * 96, hipMemcpyHostToDevice, st.s));
    launch_affine96_to_table(st.s, d_in.p, (uint32_t)n, ctx->d_table);
    return finish_srs_from_level0(ctx, st.s, n);
}

int kzg_srs_load_compressed(kzg_ctx* ctx, const uint8_t* compressed, size_t n, size_t* bad_index) {
    if (!ctx || !compressed || !n) return KZG_ERR_INVALID_ARG;
    if (bad_index) *bad_index = (size_t)-1;
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (ctx->multi) {
        ctx->last_error.clear();
        return multi_srs_load_compressed(ctx->multi, compressed, n, bad_index);
    }
    quiesce(ctx, lk);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = srs_prepare(ctx, n);
    if (rc) return rc;
    DevBuf d_in, d_status;
    HIP_TRY(ctx, hipMalloc(&d_in.p, n * 48));
    HIP_TRY(ctx, hipMalloc(&d_status.p, 4));
    TmpStream st;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
    HIP_TRY(ctx, hipMemcpyAsync(d_in.p, compressed, n * 48, hipMemcpyHostToDevice, st.s));
    HIP_TRY(ctx, hipMemsetAsync(d_status.p, 0xff, 4, st.s));
    launch_uncompress(st.s, d_in.p, (uint32_t)n, ctx->d_table, (uint32_t*)d_status.p);
    uint32_t status = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&status, d_status.p, 4, hipMemcpyDeviceToHost, st.s));
    HIP_TRY(ctx, hipStreamSynchronize(st.s));
    if (status != 0xffffffffu) {  // a malformed point (not compressed form, x >= p, not on the curve): as blst_p1_uncompress
        if (bad_index) *bad_index = status - 1;
        ctx->last_error = "compressed point " + std::to_string(status - 1) + " is malformed";
        hipFree(ctx->d_table);
        ctx->d_table = nullptr;
        return KZG_ERR_INVALID_ARG;
    }
    return finish_srs_from_level0(ctx, st.s, n);
}

// ---- binary SRS cache: 128-byte header + n x 96-byte affine points ----------------------------------------------
namespace {
struct SrsFileHeader {
    char magic[8];        // "KZGSRS1"
    uint64_t n;
    uint8_t first[48];    // compressed encodings of the first and the last point: fingerprint of the content
    uint8_t last[48];
    uint8_t reserved[16];
};
static_assert(sizeof(SrsFileHeader) == 128, "header layout");
const char kSrsMagic[8] = {'K', 'Z', 'G', 'S', 'R', 'S', '1', 0};

void affine96_compress(const uint64_t* xy, uint8_t out[48]) {
    hf::P1 p;
    std::memcpy(&p.x, xy, 48);
    std::memcpy(&p.y, xy + 6, 48);
    bool inf = true;
    for (int i = 0; i < 12; i++) inf = inf && xy[i] == 0;
    std::memset(&p.z, 0, 48);
    if (!inf) p.z = hf::kOne;
    hf::p1_compress(out, p);
}
}  // namespace

int kzg_srs_save(kzg_ctx* ctx, const char* path) {
    if (!ctx || !path) return KZG_ERR_INVALID_ARG;
    const size_t n = kzg_srs_len(ctx);
    if (!n) return KZG_ERR_NO_SRS;
    FILE* f = std::fopen(path, "wb");
    if (!f) return KZG_ERR_INVALID_ARG;
    SrsFileHeader h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, kSrsMagic, 8);
    h.n = n;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    const size_t chunk = 1 << 16;
    std::vector<uint64_t> p1(chunk * 18), xy(chunk * 12);
    for (size_t at = 0; at < n && ok; at += chunk) {
        const size_t take = n - at < chunk ? n - at : chunk;
        int rc = kzg_srs_read_g1(ctx, at, take, p1.data());
        if (rc != KZG_OK) {
            std::fclose(f);
            return rc;
        }
        for (size_t i = 0; i < take; i++) std::memcpy(&xy[12 * i], &p1[18 * i], 96);  // x, y of the affine blst_p1; infinity reads (0, 0)
        if (at == 0) affine96_compress(xy.data(), h.first);
        if (at + take == n) affine96_compress(&xy[12 * (take - 1)], h.last);
        ok = std::fwrite(xy.data(), 96, take, f) == take;
    }
    ok = ok && std::fseek(f, 0, SEEK_SET) == 0 && std::fwrite(&h, sizeof h, 1, f) == 1;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? KZG_OK : KZG_ERR_INVALID_ARG;
}

// on-curve check of the loaded points (host, spread over the cores): a corrupted cache must not commit silently
static bool affine96_all_on_curve(const uint64_t* xy, size_t n, size_t* bad) {
    size_t nthreads = std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 1;
    if (nthreads > 16) nthreads = 16;
    if (nthreads > n / 4096 + 1) nthreads = n / 4096 + 1;
    std::vector<size_t> first_bad(nthreads, (size_t)-1);
    hf::Fp four = hf::kOne + hf::kOne;
    four = four + four;
    auto work = [&](size_t t) {
        for (size_t i = t; i < n; i += nthreads) {
            hf::Fp x, y;
            std::memcpy(&x, xy + 12 * i, 48);
            std::memcpy(&y, xy + 12 * i + 6, 48);
            if (x.is_zero() && y.is_zero()) continue;  // infinity
            const bool ok = hf::geq(hf::kP, x) && !(x == hf::kP) && hf::geq(hf::kP, y) && !(y == hf::kP) &&
                            hf::sqr(y) == hf::sqr(x) * x + four;
            if (!ok) {
                first_bad[t] = i;
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nthreads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    size_t worst = (size_t)-1;
    for (size_t v : first_bad) worst = v < worst ? v : worst;
    if (bad) *bad = worst;
    return worst == (size_t)-1;
}

int kzg_srs_load_file(kzg_ctx* ctx, const char* path) {
    if (!ctx || !path) return KZG_ERR_INVALID_ARG;
    auto fail = [&](const std::string& why) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->last_error = why;
        return (int)KZG_ERR_INVALID_ARG;
    };
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail("cannot open the SRS file");
    SrsFileHeader h;
    bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, kSrsMagic, 8) == 0 && h.n > 0 && h.n <= 0x7fffffffu / 32;
    if (ok) {  // the header's n is trusted only as far as the file is that long
        ok = std::fseek(f, 0, SEEK_END) == 0;
        const long end = ok ? std::ftell(f) : -1;
        ok = ok && end >= 0 && (uint64_t)end == 128ull + 96ull * h.n && std::fseek(f, 128, SEEK_SET) == 0;
    }
    std::vector<uint64_t> xy;
    if (ok) {
        try {
            xy.resize((size_t)h.n * 12);
        } catch (const std::bad_alloc&) {  // nothing may unwind through the C-ABI
            std::fclose(f);
            return fail("out of host memory for the SRS file");
        }
        ok = std::fread(xy.data(), 96, h.n, f) == h.n;
    }
    std::fclose(f);
    if (ok) {  // fingerprint: first and last point as the header recorded them
        uint8_t a[48], b[48];
        affine96_compress(xy.data(), a);
        affine96_compress(&xy[12 * ((size_t)h.n - 1)], b);
        ok = std::memcmp(a, h.first, 48) == 0 && std::memcmp(b, h.last, 48) == 0;
    }
    if (!ok) return fail("not a KZGSRS1 file, truncated, or its fingerprint does not match its content");
    size_t bad = 0;
    if (!affine96_all_on_curve(xy.data(), (size_t)h.n, &bad)) return fail("SRS file: point " + std::to_string(bad) + " is not on the curve");
    return kzg_srs_load_affine(ctx, xy.data(), (size_t)h.n);
}

int kzg_srs_read_g1(kzg_ctx* ctx, size_t index, size_t count, uint64_t* out_p1) {
    if (!ctx || !out_p1) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->multi) return multi_srs_len(ctx->multi) ? multi_srs_read(ctx->multi, index, count, out_p1) : KZG_ERR_NO_SRS;
    if (!ctx->n) return KZG_ERR_NO_SRS;
    if (index > ctx->n || count > ctx->n - index) return KZG_ERR_INVALID_ARG;
    if (!count) return KZG_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf d_p1;
    HIP_TRY(ctx, hipMalloc(&d_p1.p, count * 144));
    hipStream_t st = ctx->slots[0].stream;
    launch_affine_to_p1(st, (const char*)ctx->d_table + index * kAffineBytes, (uint32_t)count, d_p1.p);
    HIP_TRY(ctx, hipMemcpyAsync(out_p1, d_p1.p, count * 144, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return KZG_OK;
}

// ---- submit / wait ---------------------------------------------------------------------------

// owned: the caller holds the slot through reserve_slot (SLOT_RESERVED) instead of finding it idle
static int submit_commit_locked(kzg_ctx* ctx, int slot, const uint32_t* d_scalars, int is_mont, size_t n,
                                bool tail_already_checked, bool owned = false) {
    if (!ctx->n || !ctx->slots_ready) return KZG_ERR_NO_SRS;
    if (slot < 0 || slot >= kNumSlots) return KZG_ERR_INVALID_ARG;
    Slot& s = ctx->slots[slot];
    if (s.kind != (owned ? SLOT_RESERVED : SLOT_IDLE)) return KZG_ERR_BUSY;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    s.timing = ctx->timing;
    s.has_quotient = false;
    s.job_n = n;
    s.job_batch = 1;
    s.tail_checked = tail_already_checked;
    std::memset(&s.times, 0, sizeof s.times);
    size_t n_msm = n < ctx->n ? n : ctx->n;
    std::memset(s.h_small, 0, 64 * 4);  // (the slot is idle: nothing in flight writes them)
    if (n > ctx->n && !tail_already_checked) {
        // reference: the Polynomial was truncated at construction (src/polynomial.rs:55-75), so only a
        // non-zero coefficient beyond the SRS makes the degree too high (src/polynomial.rs:201-205)
        uint64_t cnt = n - ctx->n;
        hipLaunchKernelGGL(k_tail_nonzero, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s.stream, d_scalars,
                           (uint64_t)ctx->n, (uint64_t)n, s.d_small + 24);
    }
    if (n_msm == 0) {
        s.kind = SLOT_TRIVIAL;
        return KZG_OK;
    }
    int rc = enqueue_msm(ctx, s, d_scalars, is_mont, n_msm, 0);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(s.done, s.stream));
    s.kind = SLOT_COMMIT;
    return KZG_OK;
}

int kzg_commit_submit(kzg_ctx* ctx, int slot, const void* d_coeffs, size_t n) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || (!d_coeffs && n)) return KZG_ERR_INVALID_ARG;
    if (n > kMaxCoefficients) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return submit_commit_locked(ctx, slot, (const uint32_t*)d_coeffs, 1, n, false);
}

static int submit_open_locked(kzg_ctx* ctx, int slot, const uint32_t* d_coeffs, size_t n, const uint64_t z[4],
                              const uint64_t y[4], bool owned = false) {
    if (!ctx->n || !ctx->slots_ready) return KZG_ERR_NO_SRS;
    if (slot < 0 || slot >= kNumSlots) return KZG_ERR_INVALID_ARG;
    Slot& s = ctx->slots[slot];
    if (s.kind != (owned ? SLOT_RESERVED : SLOT_IDLE)) return KZG_ERR_BUSY;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_poly(ctx, s, n);
    if (rc) return rc;
    s.timing = ctx->timing;
    s.job_n = n;
    s.job_batch = 1;
    s.has_quotient = true;
    s.tail_checked = false;
    std::memcpy(s.open_y, y, 32);
    std::memset(&s.times, 0, sizeof s.times);
    if (n == 0) {
        std::memset(s.h_small, 0, 64 * 4);
        s.kind = SLOT_TRIVIAL;
        return KZG_OK;
    }
    uint32_t zw[8];
    std::memcpy(zw, z, 32);
    if (s.timing) HIP_TRY(ctx, hipEventRecord(s.ev[6], s.stream));
    // small polynomials: one launch for the scan, the flag words and c0; otherwise memset + two launches (c0 written by the first)
    if (!launch_quotient_single(s.stream, d_coeffs, (uint32_t)n, zw, n > 1 ? s.d_q : nullptr, s.d_small)) {
        std::memset(s.h_small, 0, 64 * 4);
        PolyScratch sc{s.d_chunk, s.d_block, s.d_small, s.d_small + 8};
        launch_quotient(s.stream, d_coeffs, (uint32_t)n, zw, n > 1 ? s.d_q : nullptr, sc);
    }
    if (s.timing) HIP_TRY(ctx, hipEventRecord(s.ev[7], s.stream));
    size_t nq = n - 1;
    if (nq > ctx->n) {
        // quotient longer than the SRS: too high iff some coefficient with index > srs_len is non-zero
        uint64_t from = ctx->n + 1, cnt = n - from;
        hipLaunchKernelGGL(k_tail_nonzero, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s.stream, d_coeffs, from,
                           (uint64_t)n, s.d_small + 24);
        nq = ctx->n;
    }
    if (nq > 0) {
        rc = enqueue_msm(ctx, s, s.d_q, 1, nq, 0);
        if (rc) return rc;
    }
    HIP_TRY(ctx, hipEventRecord(s.done, s.stream));
    s.kind = nq > 0 ? SLOT_OPEN : SLOT_TRIVIAL;
    return KZG_OK;
}

int kzg_open_submit(kzg_ctx* ctx, int slot, const void* d_coeffs, size_t n, const uint64_t z[4], const uint64_t y[4]) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || !z || !y || (!d_coeffs && n)) return KZG_ERR_INVALID_ARG;
    if (n > kMaxCoefficients) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return submit_open_locked(ctx, slot, (const uint32_t*)d_coeffs, n, z, y);
}

static int wait_locked(kzg_ctx* ctx, int slot, uint64_t out_p1[18]) {
    if (slot < 0 || slot >= kNumSlots) return KZG_ERR_INVALID_ARG;
    Slot& s = ctx->slots[slot];
    if (s.kind == SLOT_IDLE) return KZG_ERR_INVALID_ARG;
    if (s.kind == SLOT_COMMIT_BATCH || s.kind == SLOT_OPEN_BATCH) {
        ctx->last_error = "kzg_wait on a slot that holds a batched job (use kzg_wait_batch / kzg_wait_open_batch)";
        return KZG_ERR_INVALID_ARG;  // the job stays in the slot
    }
    if (s.kind == SLOT_RESERVED) return KZG_ERR_BUSY;  // a synchronous call on another thread owns it
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    SlotKind kind = s.kind;
    slot_idle(ctx, s);
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    if (kind != SLOT_TRIVIAL) fill_device_times(s);
    if (s.timing && kind != SLOT_TRIVIAL) {
        float ms = 0;
        hipEventElapsedTime(&ms, s.ev[0], s.ev[1]); s.times.digits_ms = ms;
        hipEventElapsedTime(&ms, s.ev[2], s.ev[3]); s.times.scatter_ms = ms;
        fill_accumulate_times(s);
        hipEventElapsedTime(&ms, s.ev[4], s.ev[5]); s.times.reduce_ms = ms;
        if (s.has_quotient) {
            hipEventElapsedTime(&ms, s.ev[6], s.ev[7]); s.times.quotient_ms = ms;
            hipEventElapsedTime(&ms, s.ev[6], s.ev[5]); s.times.total_ms = ms;
        } else {
            hipEventElapsedTime(&ms, s.ev[0], s.ev[5]); s.times.total_ms = ms;
        }
    }
    const uint32_t* hs = s.h_small;
    hf::P1 inf = hf::p1_inf();
    if (s.has_quotient) {
        // reference order: sub -> divide_by_root (its two errors) -> commit (degree error)
        const size_t n = s.job_n;
        if (n == 0) {  // [] - [y]  (src/polynomial.rs:138-143)
            bool y_zero = true;
            for (int i = 0; i < 8; i++) y_zero &= s.open_y[i] == 0;
            if (!y_zero) return KZG_ERR_CONSTANT_POLY;
            write_p1(out_p1, inf);
            return KZG_OK;
        }
        bool higher_nonzero = hs[0] & 1u;
        if (!higher_nonzero) {  // constant polynomial after truncation (src/polynomial.rs:159-167)
            if (std::memcmp(hs + 16, s.open_y, 32) != 0) return KZG_ERR_CONSTANT_POLY;
            write_p1(out_p1, inf);
            return KZG_OK;
        }
        if (std::memcmp(hs + 8, s.open_y, 32) != 0) return KZG_ERR_REMAINDER;  // P(z) != y
        if (hs[24]) return KZG_ERR_DEGREE_TOO_HIGH;
    } else {
        if (hs[24]) return KZG_ERR_DEGREE_TOO_HIGH;
    }
    if (kind == SLOT_TRIVIAL) {
        write_p1(out_p1, inf);
        return KZG_OK;
    }
    write_p1(out_p1, finish_msm(ctx, s));
    return KZG_OK;
}

int kzg_set_max_batch(kzg_ctx* ctx, size_t max_batch) {
    if (!ctx || max_batch == 0) return KZG_ERR_INVALID_ARG;
    if (ctx->multi) return multi_set_max_batch(ctx->multi, max_batch);  // every device of the context
    std::unique_lock<std::mutex> lk(ctx->mu);
    quiesce(ctx, lk);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = drain_all(ctx);
    if (rc) return rc;
    for (auto& s : ctx->slots)
        if (s.kind != SLOT_IDLE) return KZG_ERR_BUSY;
    const uint32_t previous = ctx->max_batch;
    ctx->max_batch = (uint32_t)(max_batch > 1024 ? 1024 : max_batch);
    if (!ctx->n) return KZG_OK;  // workspaces are (re)sized when an SRS is resident
    rc = setup_slots(ctx, true);
    if (rc != KZG_OK) {
        // e.g. out of HBM: go back to the size that worked; if even that fails the SRS is dropped (KZG_ERR_NO_SRS next)
        const std::string why = ctx->last_error;
        ctx->max_batch = previous;
        (void)hipGetLastError();
        (void)setup_slots(ctx);
        ctx->last_error = why;
    }
    return rc;
}

size_t kzg_max_batch(const kzg_ctx* ctx) {
    if (ctx && ctx->multi) return kzg_max_batch(multi_kid(ctx->multi, 0));
    return ctx ? ctx->max_batch : 0;
}

static int commit_batch_submit_locked(kzg_ctx* ctx, int slot, const void* d_coeffs, size_t n, size_t batch,
                                      size_t stride_coeffs, bool owned = false) {
    if (!ctx->n || !ctx->slots_ready) return KZG_ERR_NO_SRS;
    if (slot < 0 || slot >= kNumSlots || batch > ctx->max_batch) return KZG_ERR_INVALID_ARG;
    if (n > ctx->n) return KZG_ERR_DEGREE_TOO_HIGH;  // batches take truncated polynomials only
    Slot& s = ctx->slots[slot];
    if (s.kind != (owned ? SLOT_RESERVED : SLOT_IDLE)) return KZG_ERR_BUSY;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    s.timing = ctx->timing;
    s.has_quotient = false;
    s.job_n = n;
    s.job_batch = (uint32_t)batch;
    s.tail_checked = true;
    std::memset(&s.times, 0, sizeof s.times);
    std::memset(s.h_small, 0, 64 * 4);
    int rc = enqueue_msm(ctx, s, (const uint32_t*)d_coeffs, 1, n, 0, (uint32_t)batch, stride_coeffs);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(s.done, s.stream));
    s.kind = SLOT_COMMIT_BATCH;
    return KZG_OK;
}

int kzg_commit_batch_submit(kzg_ctx* ctx, int slot, const void* d_coeffs, size_t n, size_t batch,
                            size_t stride_coeffs) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || !d_coeffs || n == 0 || batch == 0 || stride_coeffs < n || n > kMaxCoefficients) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return commit_batch_submit_locked(ctx, slot, d_coeffs, n, batch, stride_coeffs);
}

// out_stride: u64 words between consecutive results (18 = packed)
static int wait_batch_locked(kzg_ctx* ctx, int slot, uint64_t* out_p1s, size_t batch, size_t out_stride = 18) {
    if (slot < 0 || slot >= kNumSlots) return KZG_ERR_INVALID_ARG;
    Slot& s = ctx->slots[slot];
    if (s.kind != SLOT_COMMIT_BATCH || s.job_batch != batch) return KZG_ERR_INVALID_ARG;  // the job stays in the slot
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    slot_idle(ctx, s);
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    fill_device_times(s);
    if (s.timing) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, s.ev[0], s.ev[1]); s.times.digits_ms = ms;
        (void)hipEventElapsedTime(&ms, s.ev[2], s.ev[3]); s.times.scatter_ms = ms;
        fill_accumulate_times(s);
        (void)hipEventElapsedTime(&ms, s.ev[4], s.ev[5]); s.times.reduce_ms = ms;
        (void)hipEventElapsedTime(&ms, s.ev[0], s.ev[5]); s.times.total_ms = ms;
    }
    // host tails of the batch in parallel (each is ~150 point operations and one inversion)
    const uint32_t B = s.job_batch;
    const uint32_t nthreads = B <= 2 ? 1 : (B < 8 ? B : 8);  // (a thread costs ~50 us to start, a tail ~100)
    auto work = [&](uint32_t t) {
        for (uint32_t p = t; p < B; p += nthreads) write_p1(out_p1s + out_stride * (size_t)p, finish_msm(ctx, s, p, B));
    };
    if (nthreads <= 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < nthreads; t++) pool.emplace_back(work, t);
        work(0);
        for (auto& th : pool) th.join();
    }
    return KZG_OK;
}

int kzg_wait_batch(kzg_ctx* ctx, int slot, uint64_t* out_p1s, size_t batch) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || !out_p1s) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return wait_batch_locked(ctx, slot, out_p1s, batch);
}

// Batched Evaluation::generate_proof: one quotient scan per polynomial (each with its own z, y), then
// ONE batched MSM over the `batch` quotients.
static int open_batch_submit_locked(kzg_ctx* ctx, int slot, const void* d_coeffs, size_t n, size_t batch, size_t stride_coeffs,
                                    const uint64_t* zs, const uint64_t* ys, bool owned = false) {
    if (!ctx->n || !ctx->slots_ready) return KZG_ERR_NO_SRS;
    if (slot < 0 || slot >= kNumSlots || batch > ctx->max_batch) return KZG_ERR_INVALID_ARG;
    if (n - 1 > ctx->n) return KZG_ERR_DEGREE_TOO_HIGH;  // batches take truncated polynomials only
    Slot& s = ctx->slots[slot];
    if (s.kind != (owned ? SLOT_RESERVED : SLOT_IDLE)) return KZG_ERR_BUSY;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // (an owned slot's coefficients sit in its own staging buffer, which ensure_poly must not reallocate now: the
    // owner sized it before the upload)
    int rc = owned ? KZG_OK : ensure_poly(ctx, s, n * batch);
    if (rc) return rc;
    if (s.bsmall_cap < batch) {
        HIP_TRY(ctx, hipStreamSynchronize(s.stream));
        if (s.h_bsmall) hipHostFree(s.h_bsmall);
        s.h_bsmall = nullptr;
        HIP_TRY(ctx, hipHostMalloc(&s.h_bsmall, batch * 32 * 4, hipHostMallocMapped));
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&s.d_bsmall, s.h_bsmall, 0));
        s.bsmall_cap = batch;
    }
    s.timing = ctx->timing;
    s.job_n = n;
    s.job_batch = (uint32_t)batch;
    s.has_quotient = true;
    s.tail_checked = true;
    s.open_ys.assign((const uint32_t*)ys, (const uint32_t*)ys + 8 * batch);
    std::memset(&s.times, 0, sizeof s.times);
    std::memset(s.h_small, 0, 64 * 4);
    std::memset(s.h_bsmall, 0, batch * 32 * 4);
    if (s.timing) HIP_TRY(ctx, hipEventRecord(s.ev[6], s.stream));
    const size_t nq = n - 1;
    for (size_t p = 0; p < batch; p++) {
        const uint32_t* cp = (const uint32_t*)d_coeffs + p * stride_coeffs * 8;
        uint32_t zw[8];
        std::memcpy(zw, zs + 4 * p, 32);
        uint32_t* sm = s.d_bsmall + p * 32;
        PolyScratch sc{s.d_chunk, s.d_block, sm, sm + 8};  // scratch re-used in stream order
        launch_quotient(s.stream, cp, (uint32_t)n, zw, s.d_q + p * nq * 8, sc);
    }
    if (s.timing) HIP_TRY(ctx, hipEventRecord(s.ev[7], s.stream));
    rc = enqueue_msm(ctx, s, s.d_q, 1, nq, 0, (uint32_t)batch, nq);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(s.done, s.stream));
    s.kind = SLOT_OPEN_BATCH;
    return KZG_OK;
}

int kzg_open_batch_submit(kzg_ctx* ctx, int slot, const void* d_coeffs, size_t n, size_t batch, size_t stride_coeffs,
                          const uint64_t* zs, const uint64_t* ys) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || !d_coeffs || !zs || !ys || n < 2 || batch == 0 || stride_coeffs < n || n > kMaxCoefficients) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return open_batch_submit_locked(ctx, slot, d_coeffs, n, batch, stride_coeffs, zs, ys);
}

static int wait_open_batch_locked(kzg_ctx* ctx, int slot, uint64_t* out_p1s, int* statuses, size_t batch, size_t out_stride = 18,
                                  size_t status_stride = 1) {
    if (slot < 0 || slot >= kNumSlots) return KZG_ERR_INVALID_ARG;
    Slot& s = ctx->slots[slot];
    if (s.kind != SLOT_OPEN_BATCH || s.job_batch != batch || s.open_ys.size() != 8 * batch)
        return KZG_ERR_INVALID_ARG;  // the job stays in the slot
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    slot_idle(ctx, s);
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    fill_device_times(s);
    if (s.timing) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, s.ev[6], s.ev[7]); s.times.quotient_ms = ms;
        (void)hipEventElapsedTime(&ms, s.ev[0], s.ev[1]); s.times.digits_ms = ms;
        fill_accumulate_times(s);
        (void)hipEventElapsedTime(&ms, s.ev[4], s.ev[5]); s.times.reduce_ms = ms;
        (void)hipEventElapsedTime(&ms, s.ev[6], s.ev[5]); s.times.total_ms = ms;
    }
    const uint32_t B = s.job_batch;
    const hf::P1 inf = hf::p1_inf();
    for (uint32_t p = 0; p < B; p++) {
        const uint32_t* hs = s.h_bsmall + p * 32;
        const uint32_t* y = s.open_ys.data() + 8 * p;
        uint64_t* out = out_p1s + out_stride * (size_t)p;
        int& status = statuses[status_stride * (size_t)p];
        // reference order: divide_by_root's two errors (src/polynomial.rs:159-167, 184-192)
        if (!(hs[0] & 1u)) {  // constant polynomial after truncation
            if (std::memcmp(hs + 16, y, 32) != 0) status = KZG_ERR_CONSTANT_POLY;
            else { status = KZG_OK; write_p1(out, inf); }
            continue;
        }
        if (std::memcmp(hs + 8, y, 32) != 0) { status = KZG_ERR_REMAINDER; continue; }
        status = KZG_OK;
        write_p1(out, finish_msm(ctx, s, p, B));
    }
    s.open_ys.clear();
    return KZG_OK;
}

int kzg_wait_open_batch(kzg_ctx* ctx, int slot, uint64_t* out_p1s, int* statuses, size_t batch) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || !out_p1s || !statuses) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return wait_open_batch_locked(ctx, slot, out_p1s, statuses, batch);
}

int kzg_wait(kzg_ctx* ctx, int slot, uint64_t out_p1[18]) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || !out_p1) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return wait_locked(ctx, slot, out_p1);
}

// ---- synchronous host-pointer entry points (what the Rust shim calls) ---------------------------
// The context mutex is held only while a step touches shared state (slot table, enqueue order), never across the
// upload or the wait: the reference's callers are threads of `cargo test` (src/lib.rs:53, 66, 91), and N of them now
// occupy N slots whose jobs pipeline exactly like explicit kzg_*_submit calls.

// pageable host memory -> the slot's staging buffer, on the slot's stream, without the mutex
static int upload_unlocked(kzg_ctx* ctx, std::unique_lock<std::mutex>& lk, Slot& s, size_t dst_coeff, const void* src, size_t coeffs) {
    if (!coeffs) return KZG_OK;
    lk.unlock();
    const hipError_t e = hipMemcpyAsync(s.d_stage + dst_coeff * 8, src, coeffs * 32, hipMemcpyHostToDevice, s.stream);
    lk.lock();
    if (e != hipSuccess) {
        ctx->last_error = std::string("hipMemcpyAsync (coefficients): ") + hipGetErrorString(e);
        return KZG_ERR_HIP;
    }
    return KZG_OK;
}
// waits for the slot's job outside the mutex (kinds that recorded `done`), then collects it
static void await_unlocked(std::unique_lock<std::mutex>& lk, Slot& s) {
    if (s.kind == SLOT_TRIVIAL || s.kind == SLOT_RESERVED || s.kind == SLOT_IDLE) return;
    lk.unlock();
    (void)hipEventSynchronize(s.done);
    lk.lock();
}

struct HostTrace {  // phase stopwatch of one synchronous call
    kzg_ctx* ctx;
    std::chrono::steady_clock::time_point t;
    explicit HostTrace(kzg_ctx* c) : ctx(c) { if (c->host_trace) t = std::chrono::steady_clock::now(); }
    void mark(int phase) {
        if (!ctx->host_trace) return;
        const auto now = std::chrono::steady_clock::now();
        ctx->trace_ns[phase] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now - t).count();
        t = now;
    }
};

static int commit_host(kzg_ctx* ctx, const void* scalars, int is_mont, size_t n, uint64_t out_p1[18]) {
    if (!ctx || !out_p1 || (!scalars && n)) return KZG_ERR_INVALID_ARG;
    if (n > kMaxCoefficients) return KZG_ERR_INVALID_ARG;
    if (ctx->multi) return multi_commit(ctx->multi, scalars, is_mont, n, out_p1);
    HostTrace tr(ctx);
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (!ctx->n || !ctx->slots_ready) return KZG_ERR_NO_SRS;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n > ctx->n && host_tail_nonzero((const uint64_t*)scalars, ctx->n, n)) return KZG_ERR_DEGREE_TOO_HIGH;
    const int slot = reserve_slot(ctx, lk, true);
    if (slot < 0) return KZG_ERR_BUSY;
    Slot& s = ctx->slots[slot];
    const size_t n_dev = n < ctx->n ? n : ctx->n;
    int rc = ensure_poly(ctx, s, n_dev);
    tr.mark(0);
    if (rc == KZG_OK) rc = upload_unlocked(ctx, lk, s, 0, scalars, n_dev);
    tr.mark(1);
    if (rc == KZG_OK) rc = submit_commit_locked(ctx, slot, s.d_stage, is_mont, n_dev, true, true);
    tr.mark(2);
    if (rc == KZG_OK) {
        await_unlocked(lk, s);
        tr.mark(3);
        rc = wait_locked(ctx, slot, out_p1);
        tr.mark(4);
    }
    release_owned(ctx, slot);
    ctx->trace_calls++;
    return rc;
}

int kzg_commit(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, uint64_t out_p1[18]) {
    return commit_host(ctx, coeffs, 1, n, out_p1);
}
int kzg_commit_le_bytes(kzg_ctx* ctx, const uint8_t* scalars_le, size_t n, uint64_t out_p1[18]) {
    return commit_host(ctx, scalars_le, 0, n, out_p1);
}

int kzg_open(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, const uint64_t z[4], const uint64_t y[4],
             uint64_t out_p1[18]) {
    if (!ctx || !out_p1 || !z || !y || (!coeffs && n)) return KZG_ERR_INVALID_ARG;
    if (n > kMaxCoefficients) return KZG_ERR_INVALID_ARG;
    if (ctx->multi) return multi_open(ctx->multi, coeffs, n, z, y, out_p1);
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (!ctx->n || !ctx->slots_ready) return KZG_ERR_NO_SRS;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slot = reserve_slot(ctx, lk, true);
    if (slot < 0) return KZG_ERR_BUSY;
    Slot& s = ctx->slots[slot];
    int rc = ensure_poly(ctx, s, n);
    if (rc == KZG_OK) rc = upload_unlocked(ctx, lk, s, 0, coeffs, n);
    if (rc == KZG_OK) rc = submit_open_locked(ctx, slot, s.d_stage, n, z, y, true);
    if (rc == KZG_OK) {
        await_unlocked(lk, s);
        rc = wait_locked(ctx, slot, out_p1);
    }
    release_owned(ctx, slot);
    return rc;
}

// ---- host-pointer batches (BASELINE config 5: many openings against one SRS) ----------------------------------------
namespace {
struct BatchInFlight {
    int slot;
    size_t first_poly, polys;  // position in the caller's (first, step, count) sequence
};
// polynomials per submit: four or more sub-batches per call, so that the upload of one overlaps the kernels of the
// previous ones (a degree-2^20 polynomial is 32 MiB of pageable host memory), each at most max_batch polynomials -- and
// no more than ~2^19 terms: large polynomials go through the slots one by one, which is how the kernels of one overlap
// the accumulation of another (8 degree-2^20 openings per call: 378-382 /s in eight parts, 360 in four, 333 in two)
size_t host_batch_chunk(const kzg_ctx* ctx, size_t count, size_t n) {
    size_t chunk = (count + 3) / 4;
    const size_t by_size = n >= ((size_t)1 << 19) ? 1 : ((size_t)1 << 19) / (n ? n : 1);
    if (chunk > by_size) chunk = by_size;
    if (chunk > ctx->max_batch) chunk = ctx->max_batch;
    return chunk < 1 ? 1 : chunk;
}
}  // namespace

}  // extern "C"

namespace kzg {

void ctx_set_raw_partials(kzg_ctx* ctx, bool raw) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->raw_partials = raw;
}

// shared body of the two host-pointer batches: zs == nullptr -> commitments
static int batch_host(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, size_t stride, size_t first, size_t step, size_t count,
                      const uint64_t* zs, const uint64_t* ys, uint64_t* out_p1s, int* statuses) {
    if (!count) return KZG_OK;
    const bool opening = zs != nullptr;
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (!ctx->n || !ctx->slots_ready) return KZG_ERR_NO_SRS;
    if ((opening ? n - 1 : n) > ctx->n) return KZG_ERR_DEGREE_TOO_HIGH;  // batches take truncated polynomials only
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t chunk = host_batch_chunk(ctx, count, n);
    std::deque<BatchInFlight> fifo;
    auto collect_oldest = [&]() -> int {
        const BatchInFlight b = fifo.front();
        fifo.pop_front();
        Slot& s = ctx->slots[b.slot];
        await_unlocked(lk, s);
        uint64_t* out = out_p1s + 18 * (first + b.first_poly * step);
        const int r = opening ? wait_open_batch_locked(ctx, b.slot, out, statuses + (first + b.first_poly * step), b.polys, 18 * step, step)
                              : wait_batch_locked(ctx, b.slot, out, b.polys, 18 * step);
        release_owned(ctx, b.slot);
        return r;
    };
    int rc = KZG_OK;
    std::vector<uint64_t> zc, yc;
    for (size_t at = 0; at < count && rc == KZG_OK; at += chunk) {
        const size_t polys = count - at < chunk ? count - at : chunk;
        int slot = reserve_slot(ctx, lk, false);
        while (slot < 0 && rc == KZG_OK) {
            if (!fifo.empty()) rc = collect_oldest();           // my own oldest sub-batch frees a slot
            else if ((slot = reserve_slot(ctx, lk, true)) < 0) rc = KZG_ERR_BUSY;  // other callers hold them all
            if (slot < 0 && rc == KZG_OK) slot = reserve_slot(ctx, lk, false);
        }
        if (rc != KZG_OK) {
            if (slot >= 0) release_owned(ctx, slot);
            break;
        }
        Slot& s = ctx->slots[slot];
        rc = ensure_poly(ctx, s, n * polys);
        for (size_t q = 0; q < polys && rc == KZG_OK; q++)
            rc = upload_unlocked(ctx, lk, s, q * n, coeffs + (first + (at + q) * step) * stride * 4, n);
        if (rc == KZG_OK) {
            if (opening) {
                zc.resize(4 * polys);
                yc.resize(4 * polys);
                for (size_t q = 0; q < polys; q++) {
                    std::memcpy(&zc[4 * q], zs + 4 * (first + (at + q) * step), 32);
                    std::memcpy(&yc[4 * q], ys + 4 * (first + (at + q) * step), 32);
                }
                rc = open_batch_submit_locked(ctx, slot, s.d_stage, n, polys, n, zc.data(), yc.data(), true);
            } else {
                rc = commit_batch_submit_locked(ctx, slot, s.d_stage, n, polys, n, true);
            }
        }
        if (rc != KZG_OK) {
            release_owned(ctx, slot);
            break;
        }
        fifo.push_back({slot, at, polys});
    }
    while (!fifo.empty()) {
        const int r = collect_oldest();
        if (rc == KZG_OK) rc = r;
    }
    return rc;
}

int ctx_commit_batch_host(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, size_t stride_coeffs, size_t first, size_t step,
                          size_t count, uint64_t* out_p1s) {
    return batch_host(ctx, coeffs, n, stride_coeffs, first, step, count, nullptr, nullptr, out_p1s, nullptr);
}
int ctx_open_batch_host(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, size_t stride_coeffs, size_t first, size_t step,
                        size_t count, const uint64_t* zs, const uint64_t* ys, uint64_t* out_p1s, int* statuses) {
    return batch_host(ctx, coeffs, n, stride_coeffs, first, step, count, zs, ys, out_p1s, statuses);
}

// ---- one device's share of a range-sharded opening: the slice is staged once (multi.hip) ----------------------------
int ctx_open_slice_begin(kzg_ctx* ctx, const uint64_t* slice, size_t len, const uint64_t z[4], uint64_t out_h[4], int* slot_out) {
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (!ctx->n || !ctx->slots_ready) return KZG_ERR_NO_SRS;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int slot = reserve_slot(ctx, lk, true);
    if (slot < 0) return KZG_ERR_BUSY;
    Slot& s = ctx->slots[slot];
    int rc = ensure_poly(ctx, s, len + 1);
    if (rc == KZG_OK) rc = upload_unlocked(ctx, lk, s, 0, slice, len);
    if (rc == KZG_OK) {
        lk.unlock();
        uint32_t zw[8];
        std::memcpy(zw, z, 32);
        std::memset(s.h_small, 0, 64 * 4);
        hipError_t e = hipSuccess;
        PolyScratch sc{s.d_chunk, s.d_block, s.d_small, s.d_small + 8};
        launch_quotient(s.stream, s.d_stage, (uint32_t)len, zw, nullptr, sc);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
        lk.lock();
        if (e != hipSuccess) {
            ctx->last_error = std::string("slice evaluation: ") + hipGetErrorString(e);
            rc = KZG_ERR_HIP;
        }
    }
    if (rc != KZG_OK) {
        release_owned(ctx, slot);
        return rc;
    }
    std::memcpy(out_h, s.h_small + 8, 32);
    *slot_out = slot;
    return KZG_OK;
}

int ctx_open_slice_finish(kzg_ctx* ctx, int slot, size_t len, const uint64_t carry[4], const uint64_t z[4],
                          const uint64_t start[4], uint64_t out_p1[18]) {
    std::unique_lock<std::mutex> lk(ctx->mu);
    Slot& s = ctx->slots[slot];
    int rc = KZG_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) rc = KZG_ERR_HIP;
    if (rc == KZG_OK) rc = upload_unlocked(ctx, lk, s, len, carry, 1);  // the carry as one more top coefficient
    if (rc == KZG_OK) rc = submit_open_locked(ctx, slot, s.d_stage, len + 1, z, start, true);
    if (rc == KZG_OK) {
        await_unlocked(lk, s);
        rc = wait_locked(ctx, slot, out_p1);
    }
    release_owned(ctx, slot);
    return rc;
}

void ctx_open_slice_abort(kzg_ctx* ctx, int slot) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    release_owned(ctx, slot);
}

}  // namespace kzg

extern "C" {

int kzg_commit_batch(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, size_t batch, size_t stride_coeffs, uint64_t* out_p1s) {
    if (!ctx || (!coeffs && batch) || (!out_p1s && batch) || n == 0 || stride_coeffs < n || n > kMaxCoefficients) return KZG_ERR_INVALID_ARG;
    if (ctx->multi) return multi_commit_batch(ctx->multi, coeffs, n, batch, stride_coeffs, out_p1s);
    return ctx_commit_batch_host(ctx, coeffs, n, stride_coeffs, 0, 1, batch, out_p1s);
}

int kzg_open_batch(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, size_t batch, size_t stride_coeffs, const uint64_t* zs,
                   const uint64_t* ys, uint64_t* out_p1s, int* statuses) {
    if (!ctx || ((!coeffs || !zs || !ys || !out_p1s || !statuses) && batch) || stride_coeffs < n || n > kMaxCoefficients)
        return KZG_ERR_INVALID_ARG;
    if (n < 2) {  // empty and constant polynomials: the single-opening path knows their rules (src/polynomial.rs:138-167)
        for (size_t p = 0; p < batch; p++) {
            const int rc = kzg_open(ctx, coeffs + p * stride_coeffs * 4, n, zs + 4 * p, ys + 4 * p, out_p1s + 18 * p);
            if (rc != KZG_OK && rc != KZG_ERR_CONSTANT_POLY && rc != KZG_ERR_REMAINDER && rc != KZG_ERR_DEGREE_TOO_HIGH) return rc;
            statuses[p] = rc;
        }
        return KZG_OK;
    }
    if (ctx->multi) return multi_open_batch(ctx->multi, coeffs, n, batch, stride_coeffs, zs, ys, out_p1s, statuses);
    return ctx_open_batch_host(ctx, coeffs, n, stride_coeffs, 0, 1, batch, zs, ys, out_p1s, statuses);
}

int kzg_quotient(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, const uint64_t z[4], const uint64_t y[4],
                 uint64_t* out_q, size_t* out_qn) {
    if (!ctx || !z || !y || !out_qn || (!coeffs && n) || (!out_q && n > 1)) return KZG_ERR_INVALID_ARG;
    if (n > kMaxCoefficients) return KZG_ERR_INVALID_ARG;
    if (ctx->multi) return kzg_quotient(multi_kid(ctx->multi, 0), coeffs, n, z, y, out_q, out_qn);  // needs no SRS
    std::unique_lock<std::mutex> lk(ctx->mu);
    *out_qn = 0;
    const int slot = reserve_slot(ctx, lk, true);
    if (slot < 0) return KZG_ERR_BUSY;
    SlotLease lease{ctx, slot};
    Slot& s = ctx->slots[slot];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rcb = ensure_slot_basics(ctx, s);  // quotient does not need an SRS
    if (rcb) return rcb;
    bool y_zero = (y[0] | y[1] | y[2] | y[3]) == 0;
    if (n == 0) return y_zero ? KZG_OK : KZG_ERR_CONSTANT_POLY;
    int rc = ensure_poly(ctx, s, n);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(s.d_stage, coeffs, n * 32, hipMemcpyHostToDevice, s.stream));
    std::memset(s.h_small, 0, 64 * 4);
    uint32_t zw[8];
    std::memcpy(zw, z, 32);
    PolyScratch sc{s.d_chunk, s.d_block, s.d_small, s.d_small + 8};
    launch_quotient(s.stream, s.d_stage, (uint32_t)n, zw, n > 1 ? s.d_q : nullptr, sc);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    bool higher_nonzero = s.h_small[0] & 1u;
    if (!higher_nonzero) return std::memcmp(coeffs, y, 32) == 0 ? KZG_OK : KZG_ERR_CONSTANT_POLY;
    if (std::memcmp(s.h_small + 8, y, 32) != 0) return KZG_ERR_REMAINDER;
    // quotient of the truncated polynomial: its length is (index of the last non-zero coefficient)
    size_t n_eff = n;
    while (n_eff > 1 && !(coeffs[4 * (n_eff - 1)] | coeffs[4 * (n_eff - 1) + 1] | coeffs[4 * (n_eff - 1) + 2] |
                          coeffs[4 * (n_eff - 1) + 3]))
        n_eff--;
    HIP_TRY(ctx, hipMemcpy(out_q, s.d_q, (n_eff - 1) * 32, hipMemcpyDeviceToHost));
    *out_qn = n_eff - 1;
    return KZG_OK;
}

int kzg_evaluate(kzg_ctx* ctx, const uint64_t* coeffs, size_t n, const uint64_t z[4], uint64_t out_y[4]) {
    if (!ctx || !z || !out_y || (!coeffs && n)) return KZG_ERR_INVALID_ARG;
    if (n > kMaxCoefficients) return KZG_ERR_INVALID_ARG;
    if (ctx->multi) return kzg_evaluate(multi_kid(ctx->multi, 0), coeffs, n, z, out_y);  // needs no SRS
    std::unique_lock<std::mutex> lk(ctx->mu);
    std::memset(out_y, 0, 32);
    if (n == 0) return KZG_OK;
    const int slot = reserve_slot(ctx, lk, true);
    if (slot < 0) return KZG_ERR_BUSY;
    SlotLease lease{ctx, slot};
    Slot& s = ctx->slots[slot];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rcb = ensure_slot_basics(ctx, s);
    if (rcb) return rcb;
    int rc = ensure_poly(ctx, s, n);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(s.d_stage, coeffs, n * 32, hipMemcpyHostToDevice, s.stream));
    std::memset(s.h_small, 0, 64 * 4);
    uint32_t zw[8];
    std::memcpy(zw, z, 32);
    PolyScratch sc{s.d_chunk, s.d_block, s.d_small, s.d_small + 8};
    launch_quotient(s.stream, s.d_stage, (uint32_t)n, zw, nullptr, sc);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s.stream));
    std::memcpy(out_y, s.h_small + 8, 32);
    return KZG_OK;
}

// ---- raw device memory -----------------------------------------------------------------------

int kzg_dev_alloc(kzg_ctx* ctx, size_t bytes, void** out) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || !out) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(out, bytes ? bytes : 1));
    return KZG_OK;
}
int kzg_dev_free(kzg_ctx* ctx, void* p) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipFree(p));
    return KZG_OK;
}
int kzg_dev_upload(kzg_ctx* ctx, void* dst, const void* src, size_t bytes) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || (!dst && bytes) || (!src && bytes)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return KZG_OK;
}
int kzg_dev_download(kzg_ctx* ctx, void* dst, const void* src, size_t bytes) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || (!dst && bytes) || (!src && bytes)) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return KZG_OK;
}

// ---- host-side G1 helpers ----------------------------------------------------------------------

int kzg_g1_sum(const uint64_t* p1s, size_t k, uint64_t out_p1[18]) {
    if (!out_p1 || (!p1s && k)) return KZG_ERR_INVALID_ARG;
    hf::P1 acc = hf::p1_inf();
    for (size_t i = 0; i < k; i++) {
        hf::P1 p;
        std::memcpy(&p, p1s + 18 * i, sizeof p);
        acc = hf::p1_add(acc, p);
    }
    write_p1(out_p1, hf::p1_normalize(acc));
    return KZG_OK;
}

int kzg_g1_compress(const uint64_t p1[18], uint8_t out[48]) {
    if (!p1 || !out) return KZG_ERR_INVALID_ARG;
    hf::P1 p;
    std::memcpy(&p, p1, sizeof p);
    hf::p1_compress(out, p);
    return KZG_OK;
}

int kzg_g1_uncompress(const uint8_t in[48], uint64_t out_p1[18]) {
    if (!in || !out_p1) return KZG_ERR_INVALID_ARG;
    hf::P1 p;
    if (!hf::p1_uncompress(p, in)) return KZG_ERR_INVALID_ARG;
    write_p1(out_p1, p);
    return KZG_OK;
}

int kzg_verify_proof(const uint64_t commitment_p1[18], const uint64_t proof_p1[18], const uint64_t z[4],
                     const uint64_t y[4], const uint64_t s_g2_p2[36], int* valid) {
    if (!commitment_p1 || !proof_p1 || !z || !y || !s_g2_p2 || !valid) return KZG_ERR_INVALID_ARG;
    int r = hf::verify_proof(commitment_p1, proof_p1, z, y, s_g2_p2);
    if (r < 0) return KZG_ERR_INVALID_ARG;
    *valid = r;
    return KZG_OK;
}

int kzg_verify_proof_batch(const uint64_t* commitments_p1, const uint64_t* proofs_p1, const uint64_t* zs,
                           const uint64_t* ys, const uint64_t s_g2_p2[36], size_t n, int* valid) {
    if ((!commitments_p1 || !proofs_p1 || !zs || !ys || !valid) && n) return KZG_ERR_INVALID_ARG;
    if (!s_g2_p2) return KZG_ERR_INVALID_ARG;
    size_t nthreads = std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 1;
    if (nthreads > n) nthreads = n;
    std::vector<int> results(n, 0);
    auto work = [&](size_t t) {
        for (size_t i = t; i < n; i += nthreads)
            results[i] = hf::verify_proof(commitments_p1 + 18 * i, proofs_p1 + 18 * i, zs + 4 * i, ys + 4 * i, s_g2_p2);
    };
    if (nthreads <= 1) {
        if (n) work(0);
    } else {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nthreads; t++) pool.emplace_back(work, t);
        work(0);
        for (auto& th : pool) th.join();
    }
    for (size_t i = 0; i < n; i++) {
        if (results[i] < 0) return KZG_ERR_INVALID_ARG;
        valid[i] = results[i];
    }
    return KZG_OK;
}

// SetupArtifact k's G2 half on the host: [s^k mod r]G2 (reference src/trusted_setup.rs:40-53 for the power, :64-72 for
// the point; s = secret read big-endian, src/trusted_setup.rs:20-28)
int kzg_srs_g2_at(const uint8_t secret_be[32], uint64_t index, uint64_t out_p2[36]) {
    if (!secret_be || !out_p2) return KZG_ERR_INVALID_ARG;
    hf::Fr s;
    for (int w = 0; w < 4; w++) {
        uint64_t v = 0;
        for (int b = 0; b < 8; b++) v = (v << 8) | secret_be[24 - 8 * w + b];
        s.l[w] = v;
    }
    uint64_t br = 0;
    while (hf::fr_geq(s, hf::kFrMod)) s = hf::fr_raw_sub(s, hf::kFrMod, br);  // 2^256 < 3 r
    static const hf::Fr kR2 = {{0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL}};
    const hf::Fr power = hf::fr_pow(hf::fr_mul(s, kR2), index);  // Montgomery form of s^index
    uint64_t e[4];
    hf::fr_from_mont(power.l, e);
    const hf::P2 q = hf::p2_normalize(hf::p2_mul(hf::p2_generator(), e));
    std::memcpy(out_p2, &q, sizeof q);
    return KZG_OK;
}

// ---- measurement ---------------------------------------------------------------------------------

int kzg_set_timing(kzg_ctx* ctx, int enabled) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->timing = enabled != 0;
    return KZG_OK;
}
int kzg_get_times(kzg_ctx* ctx, int slot, kzg_kernel_times* out) {
    KZG_SINGLE_DEVICE_ONLY(ctx);
    if (!ctx || !out || slot < 0 || slot >= kNumSlots) return KZG_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    *out = ctx->slots[slot].times;
    return KZG_OK;
}

}  // extern "C"
