// multi.hip -- a context that spans several devices of one node (kzg_ctx_create_multi / _ex, include/kzg_mi355x.h).
//
// The reference passes the whole SRS by slice to Polynomial::commit / Evaluation::generate_proof
// (src/polynomial.rs:200-215, 260-269).  Two ways of spreading that over K devices:
//
// RANGE mode (kzg_ctx_create_multi; BASELINE config 4): device g keeps SRS points [lo_g, hi_g) resident with their
// window tables and every call is sharded transparently:
//   commit  sum_i c_i SRS_i = sum_g sum_{i in slice g} c_i SRS_i: one partial MSM per device, then the exchange below;
//   open    q[j] = S[j+1] with S[i] = sum_{k>=i} c_k z^(k-i) (src/polynomial.rs:168-179): device g stages its slice ONCE
//           and evaluates it (the quotient scan without output), the host runs the K-step recurrence
//           C_{g-1} = H_g + z^(len_g) C_g (32 bytes per device), device g appends its carry C_g to the staged slice as
//           one more top coefficient and opens it; the partial proofs combine like partial commitments.  No kernel knows
//           about the sharding.
//   exchange  "RCCL reduce of Jacobian partial sums": the devices hand over UN-normalised Jacobian partials (no
//           inversion per device); RCCL has no user-defined reduction for curve points, so reduce = ncclAllGather of
//           the 144-byte blst_p1 partials (librccl linked directly, single process: ncclCommInitAll + one group call)
//           + K-1 complete additions + ONE normalisation.  The partial sums are born on the host (the last ~100
//           additions of an MSM are its host tail, api.hip), so the RCCL leg is an upload, the collective and a
//           download of 144 bytes per device: it is kept because the exchange over xGMI is what the north star asks
//           for and what a multi-process deployment (bench.py --gpus N) needs; KZG_MULTI_EXCHANGE=host adds the
//           partials where they already are.  When a device appears more than once in the list (virtual slices on one
//           GPU: how a single-GPU box rehearses the path) or RCCL cannot form the communicator, the host path is taken.
//
// REPLICATE mode (kzg_ctx_create_multi_ex with KZG_MULTI_REPLICATE_SRS; BASELINE config 5): every device keeps the whole
// SRS; polynomial p of a batch goes to device p mod K, each device pipelines its share through its stream slots
// (upload of one sub-batch under the kernels of the previous), nothing is exchanged.  Single calls take the devices in
// turn.
//
// Every device has a persistent host thread (WorkerPool): a call publishes one closure, the workers (and the caller,
// as device 0) run it and the caller collects the statuses -- no thread is created per call.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "engine.h"
#include "host_field.hpp"
#include "host_fr.hpp"

namespace hf = kzg_host;

namespace kzg {

namespace {

inline void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

// K-1 persistent threads, one per device after the first; run(fn) executes fn(g) for every g in [0, K), fn(0) on the
// calling thread.  A worker spins for 200 us after a job (the next round of the same call arrives within microseconds)
// and then sleeps on a condition variable.  (Spinning for 1.5 ms instead was measured and is worse on a box that grants
// the process 16 cores: seven spinning threads take them from the HIP runtime's own threads.)
class WorkerPool {
  public:
    WorkerPool(const std::vector<int>& devices) : k_(devices.size()), rc_(devices.size(), KZG_OK) {
        for (size_t g = 1; g < k_; g++) threads_.emplace_back([this, g, dev = devices[g]] { loop(g, dev); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_.store(true);
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    // returns the statuses in device order
    const std::vector<int>& run(const std::function<int(size_t)>& fn) {
        job_ = &fn;
        pending_.store(k_ - 1);
        gen_.fetch_add(1);
        if (sleepers_.load() > 0) {
            std::lock_guard<std::mutex> lk(mu_);
            cv_.notify_all();
        }
        rc_[0] = fn(0);
        for (unsigned spins = 0; pending_.load() != 0; spins++) {
            if (spins < 4096) cpu_relax();
            else std::this_thread::yield();
        }
        return rc_;
    }

  private:
    void loop(size_t g, int device) {
        (void)hipSetDevice(device);
        uint64_t seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (gen_.load() == seen && !stop_.load()) {
                cpu_relax();
                if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                    std::unique_lock<std::mutex> lk(mu_);
                    sleepers_.fetch_add(1);
                    cv_.wait(lk, [&] { return gen_.load() != seen || stop_.load(); });
                    sleepers_.fetch_sub(1);
                }
            }
            if (stop_.load()) return;
            seen = gen_.load();
            rc_[g] = (*job_)(g);
            pending_.fetch_sub(1);
        }
    }
    size_t k_;
    std::vector<std::thread> threads_;
    std::vector<int> rc_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<size_t> pending_{0};
    std::atomic<int> sleepers_{0};
    std::atomic<bool> stop_{false};
    const std::function<int(size_t)>* job_ = nullptr;
};

constexpr size_t kExchangeMaxPartials = 64;  // partial sums per device and exchange (a batch is exchanged in pieces of this)

}  // namespace

struct MultiState {
    uint32_t mode = kMultiRange;
    std::vector<kzg_ctx*> kids;
    std::vector<int> devices;
    std::vector<size_t> lo, hi;  // SRS range of every kid (replicate mode: [0, n) for all)
    size_t n = 0;
    std::mutex op_mu;            // range-mode calls and batches use every device: one at a time per context
    WorkerPool* pool = nullptr;
    std::atomic<uint64_t> next_kid{0};  // replicate mode: single calls take the devices in turn
    // exchange
    bool rccl = false;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<uint64_t*> d_in, d_out;  // 18 * kExchangeMaxPartials and K times that, per device
    uint64_t* h_stage = nullptr;         // pinned: K x kExchangeMaxPartials x 18 words up, the same down
    std::string last_error;
    std::atomic<uint64_t> rccl_exchanges{0};
    // KZG_HOST_TRACE=1: per sharded call, wall time of the call, of its slowest device thread and of the exchange + sum
    // (printed when the context is destroyed): call - slowest device = what the context adds on top of its shards
    bool trace = false, trace_in_call = false;
    uint64_t trace_calls = 0, trace_call_ns = 0, trace_slowest_ns = 0, trace_exchange_ns = 0;
    std::vector<uint64_t> kid_ns;  // per device, current round
};

static uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void shard_range(size_t n, size_t g, size_t k, size_t& lo, size_t& hi) {
    const size_t per = (n + k - 1) / k;
    lo = g * per < n ? g * per : n;
    hi = lo + per < n ? lo + per : n;
}

static void release_exchange(MultiState* m) {
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    for (size_t g = 0; g < m->devices.size(); g++) {
        (void)hipSetDevice(m->devices[g]);
        if (g < m->d_in.size() && m->d_in[g]) (void)hipFree(m->d_in[g]);
        if (g < m->d_out.size() && m->d_out[g]) (void)hipFree(m->d_out[g]);
        if (g < m->streams.size() && m->streams[g]) (void)hipStreamDestroy(m->streams[g]);
        if (g < m->comms.size() && m->comms[g]) (void)ncclCommDestroy(m->comms[g]);
    }
    if (m->h_stage) (void)hipHostFree(m->h_stage);
    m->d_in.clear();
    m->d_out.clear();
    m->streams.clear();
    m->comms.clear();
    m->h_stage = nullptr;
    m->rccl = false;
    if (have_prev) (void)hipSetDevice(prev);
}

int multi_create(const int* devices, int ndev, uint32_t mode, MultiState** out, std::string& err) {
    *out = nullptr;
    if (!devices || ndev <= 0) return KZG_ERR_INVALID_ARG;
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    MultiState* m = new MultiState();
    m->mode = mode;
    for (int g = 0; g < ndev; g++) {
        kzg_ctx* kid = nullptr;
        int rc = kzg_ctx_create(devices[g], &kid);
        if (rc != KZG_OK) {
            for (auto* k : m->kids) kzg_ctx_destroy(k);
            delete m;
            if (have_prev) (void)hipSetDevice(prev);
            return rc;
        }
        if (mode == kMultiRange) ctx_set_raw_partials(kid, true);
        m->kids.push_back(kid);
        m->devices.push_back(devices[g]);
    }
    m->lo.assign(ndev, 0);
    m->hi.assign(ndev, 0);
    m->pool = new WorkerPool(m->devices);
    m->kid_ns.assign(ndev, 0);
    if (const char* v = std::getenv("KZG_HOST_TRACE")) m->trace = std::atoi(v) != 0;
    // The exchange of a range-split context goes over RCCL when a communicator can be formed: it needs distinct
    // devices (KZG_MULTI_FORCE_RCCL=1 forms one for a single device too -- a world of one, which is how the GPU suite
    // exercises the exchange code on a one-GPU box).  KZG_MULTI_EXCHANGE=host never forms one.
    std::set<int> distinct(devices, devices + ndev);
    const char* force = std::getenv("KZG_MULTI_FORCE_RCCL");
    const char* how = std::getenv("KZG_MULTI_EXCHANGE");
    const bool host_only = how && !std::strcmp(how, "host");
    if (mode == kMultiRange && !host_only && (ndev > 1 || (force && force[0] == '1')) && (int)distinct.size() == ndev) {
        m->comms.assign(ndev, nullptr);
        ncclResult_t r = ncclCommInitAll(m->comms.data(), ndev, devices);
        if (r != ncclSuccess) {
            // the partial sums are on the host anyway: carry on without the collective
            err = std::string("ncclCommInitAll: ") + ncclGetErrorString(r) + " (partial sums will be added on the host)";
            m->last_error = err;
            m->comms.clear();
        } else {
            m->streams.assign(ndev, nullptr);
            m->d_in.assign(ndev, nullptr);
            m->d_out.assign(ndev, nullptr);
            bool ok = true;
            for (int g = 0; g < ndev && ok; g++) {
                ok = hipSetDevice(devices[g]) == hipSuccess &&
                     hipStreamCreateWithFlags(&m->streams[g], hipStreamNonBlocking) == hipSuccess &&
                     hipMalloc((void**)&m->d_in[g], kExchangeMaxPartials * 18 * 8) == hipSuccess &&
                     hipMalloc((void**)&m->d_out[g], (size_t)ndev * kExchangeMaxPartials * 18 * 8) == hipSuccess;
            }
            ok = ok && hipHostMalloc((void**)&m->h_stage, 2 * (size_t)ndev * kExchangeMaxPartials * 18 * 8) == hipSuccess;
            if (ok) {
                m->rccl = true;
            } else {
                (void)hipGetLastError();
                m->last_error = err = "multi-device exchange buffers could not be allocated (partial sums will be added on the host)";
                release_exchange(m);
            }
        }
    }
    if (have_prev) (void)hipSetDevice(prev);
    *out = m;
    return KZG_OK;
}

void multi_destroy(MultiState* m) {
    if (!m) return;
    if (m->trace && m->trace_calls) {
        const double k = 1e-3 / (double)m->trace_calls;
        std::fprintf(stderr, "[kzg multi trace] %llu sharded calls on %zu devices, us per call: total %.1f slowest device of every round %.1f exchange+sum %.1f "
                             "=> context overhead %.1f\n",
                     (unsigned long long)m->trace_calls, m->kids.size(), k * m->trace_call_ns, k * m->trace_slowest_ns,
                     k * m->trace_exchange_ns, k * (m->trace_call_ns - m->trace_slowest_ns));
    }
    delete m->pool;
    release_exchange(m);
    for (auto* k : m->kids) kzg_ctx_destroy(k);
    delete m;
}

size_t multi_srs_len(const MultiState* m) { return m->n; }
int multi_num_devices(const MultiState* m) { return (int)m->kids.size(); }
uint64_t multi_rccl_exchanges(const MultiState* m) { return m->rccl_exchanges.load(); }
kzg_ctx* multi_kid(MultiState* m, int g) { return m->kids[g]; }
const char* multi_last_error(const MultiState* m) { return m->last_error.c_str(); }
uint32_t multi_mode(const MultiState* m) { return m->mode; }

// fn(g) on every device, each on its own persistent host thread; returns the first non-OK status in device order
static int for_each_kid(MultiState* m, const std::function<int(size_t)>& fn) {
    const std::function<int(size_t)> traced = [&](size_t g) {
        const uint64_t t0 = now_ns();
        const int r = fn(g);
        m->kid_ns[g] += now_ns() - t0;
        return r;
    };
    if (m->trace) std::fill(m->kid_ns.begin(), m->kid_ns.end(), 0);
    const std::vector<int>& rc = m->pool->run(m->trace ? traced : fn);
    if (m->trace && m->trace_in_call) m->trace_slowest_ns += *std::max_element(m->kid_ns.begin(), m->kid_ns.end());  // per round over the devices
    for (size_t g = 0; g < rc.size(); g++)
        if (rc[g] != KZG_OK) {
            m->last_error = std::string("device slice ") + std::to_string(g) + ": " + kzg_last_error(m->kids[g]);
            return rc[g];
        }
    return KZG_OK;
}

// the slice of an n-point SRS that kid g keeps
static void assign_ranges(MultiState* m, size_t n) {
    const size_t k = m->kids.size();
    for (size_t g = 0; g < k; g++) {
        if (m->mode == kMultiReplicate) {
            m->lo[g] = 0;
            m->hi[g] = n;
        } else {
            shard_range(n, g, k, m->lo[g], m->hi[g]);
        }
    }
}

int multi_srs_generate(MultiState* m, const uint8_t secret_be[32], uint64_t first, size_t n) {
    std::lock_guard<std::mutex> lk(m->op_mu);
    m->n = 0;
    assign_ranges(m, n);
    int rc = for_each_kid(m, [&](size_t g) {
        if (m->hi[g] == m->lo[g]) return (int)KZG_OK;  // empty slice: the kid simply holds no SRS
        return kzg_srs_generate_g1(m->kids[g], secret_be, first + m->lo[g], m->hi[g] - m->lo[g]);
    });
    if (rc == KZG_OK) m->n = n;
    return rc;
}

int multi_srs_load(MultiState* m, const void* first_g1, size_t stride, size_t n) {
    std::lock_guard<std::mutex> lk(m->op_mu);
    m->n = 0;
    assign_ranges(m, n);
    int rc = for_each_kid(m, [&](size_t g) {
        if (m->hi[g] == m->lo[g]) return (int)KZG_OK;
        return kzg_srs_load_g1(m->kids[g], (const char*)first_g1 + m->lo[g] * stride, stride, m->hi[g] - m->lo[g]);
    });
    if (rc == KZG_OK) m->n = n;
    return rc;
}

int multi_srs_load_affine(MultiState* m, const void* affine_xy, size_t n) {
    std::lock_guard<std::mutex> lk(m->op_mu);
    m->n = 0;
    assign_ranges(m, n);
    int rc = for_each_kid(m, [&](size_t g) {
        if (m->hi[g] == m->lo[g]) return (int)KZG_OK;
        return kzg_srs_load_affine(m->kids[g], (const char*)affine_xy + m->lo[g] * 96, m->hi[g] - m->lo[g]);
    });
    if (rc == KZG_OK) m->n = n;
    return rc;
}

int multi_srs_load_compressed(MultiState* m, const uint8_t* compressed, size_t n, size_t* bad_index) {
    std::lock_guard<std::mutex> lk(m->op_mu);
    const size_t k = m->kids.size();
    m->n = 0;
    assign_ranges(m, n);
    std::vector<size_t> bad(k, (size_t)-1);
    int rc = for_each_kid(m, [&](size_t g) {
        if (m->hi[g] == m->lo[g]) return (int)KZG_OK;
        return kzg_srs_load_compressed(m->kids[g], compressed + m->lo[g] * 48, m->hi[g] - m->lo[g], &bad[g]);
    });
    if (bad_index)
        for (size_t g = 0; g < k; g++)
            if (bad[g] != (size_t)-1) {
                *bad_index = m->lo[g] + bad[g];
                break;
            }
    if (rc == KZG_OK) m->n = n;
    return rc;
}

int multi_srs_read(MultiState* m, size_t index, size_t count, uint64_t* out_p1) {
    if (index > m->n || count > m->n - index) return KZG_ERR_INVALID_ARG;
    if (m->mode == kMultiReplicate) return count ? kzg_srs_read_g1(m->kids[0], index, count, out_p1) : KZG_OK;
    for (size_t g = 0; g < m->kids.size() && count; g++) {
        if (index >= m->hi[g] || m->hi[g] == m->lo[g]) continue;
        const size_t take = (m->hi[g] - index < count) ? m->hi[g] - index : count;
        int rc = kzg_srs_read_g1(m->kids[g], index - m->lo[g], take, out_p1);
        if (rc != KZG_OK) return rc;
        out_p1 += 18 * take;
        index += take;
        count -= take;
    }
    return KZG_OK;
}

int multi_set_max_batch(MultiState* m, size_t max_batch) {
    std::lock_guard<std::mutex> lk(m->op_mu);
    return for_each_kid(m, [&](size_t g) { return kzg_set_max_batch(m->kids[g], max_batch); });
}

// The exchange and the sum.  partials: device-major [g][b], b < count, 18 words each (un-normalised Jacobian, all-zero
// Z = infinity); out: count normalised blst_p1.  Over RCCL when the context owns a communicator.
static int exchange_and_sum_impl(MultiState* m, const uint64_t* partials, size_t count, uint64_t* out_p1s);
static int exchange_and_sum(MultiState* m, const uint64_t* partials, size_t count, uint64_t* out_p1s) {
    const uint64_t t0 = m->trace ? now_ns() : 0;
    const int rc = exchange_and_sum_impl(m, partials, count, out_p1s);
    if (m->trace) m->trace_exchange_ns += now_ns() - t0;
    return rc;
}
static int exchange_and_sum_impl(MultiState* m, const uint64_t* partials, size_t count, uint64_t* out_p1s) {
    const size_t k = m->kids.size();
    std::vector<hf::P1> sums(count);
    if (!m->rccl) {
        for (size_t b = 0; b < count; b++) {
            hf::P1 acc = hf::p1_inf();
            for (size_t g = 0; g < k; g++) {
                hf::P1 p;
                std::memcpy(&p, partials + 18 * (g * count + b), sizeof p);
                acc = hf::p1_add(acc, p);
            }
            sums[b] = acc;
        }
    } else {
        int prev = 0;
        const bool have_prev = hipGetDevice(&prev) == hipSuccess;
        auto drain = [&] {  // nothing may still read the staging buffers or hold a stream when this function returns
            for (size_t g = 0; g < k; g++)
                if (hipSetDevice(m->devices[g]) == hipSuccess) (void)hipStreamSynchronize(m->streams[g]);
            if (have_prev) (void)hipSetDevice(prev);
        };
        uint64_t* h_up = m->h_stage;
        uint64_t* h_down = m->h_stage + k * kExchangeMaxPartials * 18;
        for (size_t b0 = 0; b0 < count; b0 += kExchangeMaxPartials) {
            const size_t nb = count - b0 < kExchangeMaxPartials ? count - b0 : kExchangeMaxPartials;
            const size_t words = 18 * nb;
            for (size_t g = 0; g < k; g++) {
                std::memcpy(h_up + g * words, partials + 18 * (g * count + b0), words * 8);
                if (hipSetDevice(m->devices[g]) != hipSuccess ||
                    hipMemcpyAsync(m->d_in[g], h_up + g * words, words * 8, hipMemcpyHostToDevice, m->streams[g]) != hipSuccess) {
                    m->last_error = "exchange: upload of a partial sum failed";
                    drain();
                    return KZG_ERR_HIP;
                }
            }
            ncclResult_t r = ncclGroupStart();
            for (size_t g = 0; g < k && r == ncclSuccess; g++)
                r = ncclAllGather(m->d_in[g], m->d_out[g], words, ncclUint64, m->comms[g], m->streams[g]);
            ncclResult_t r2 = ncclGroupEnd();
            if (r != ncclSuccess || r2 != ncclSuccess) {
                m->last_error = std::string("ncclAllGather: ") + ncclGetErrorString(r != ncclSuccess ? r : r2);
                drain();
                return KZG_ERR_HIP;
            }
            // every device now holds all K pieces; the host takes device 0's copy (and waits for every stream so that
            // the buffers can be reused)
            if (hipSetDevice(m->devices[0]) != hipSuccess ||
                hipMemcpyAsync(h_down, m->d_out[0], k * words * 8, hipMemcpyDeviceToHost, m->streams[0]) != hipSuccess) {
                m->last_error = "exchange: download of the gathered partial sums failed";
                drain();
                return KZG_ERR_HIP;
            }
            bool ok = true;
            for (size_t g = 0; g < k; g++)
                ok = (hipSetDevice(m->devices[g]) == hipSuccess && hipStreamSynchronize(m->streams[g]) == hipSuccess) && ok;
            if (!ok) {
                m->last_error = "exchange: stream synchronisation failed";
                if (have_prev) (void)hipSetDevice(prev);
                return KZG_ERR_HIP;
            }
            m->rccl_exchanges.fetch_add(1);
            for (size_t b = 0; b < nb; b++) {
                hf::P1 acc = hf::p1_inf();
                for (size_t g = 0; g < k; g++) {
                    hf::P1 p;
                    std::memcpy(&p, h_down + g * words + 18 * b, sizeof p);
                    acc = hf::p1_add(acc, p);
                }
                sums[b0 + b] = acc;
            }
        }
        if (have_prev) (void)hipSetDevice(prev);
    }
    hf::p1_normalize_many(sums.data(), count);  // ONE inversion for the call
    std::memcpy(out_p1s, sums.data(), count * sizeof(hf::P1));
    return KZG_OK;
}

struct CallTrace {  // one sharded call: started after op_mu is taken
    MultiState* m;
    uint64_t t0;
    explicit CallTrace(MultiState* ms) : m(ms), t0(ms->trace ? now_ns() : 0) { m->trace_in_call = true; }
    ~CallTrace() {
        m->trace_in_call = false;
        if (!m->trace) return;
        m->trace_calls++;
        m->trace_call_ns += now_ns() - t0;
    }
};

static bool fr_words_nonzero(const uint64_t* c, size_t from, size_t to) {
    for (size_t i = from; i < to; i++)
        if (c[4 * i] | c[4 * i + 1] | c[4 * i + 2] | c[4 * i + 3]) return true;
    return false;
}

int multi_commit(MultiState* m, const void* scalars, int is_mont, size_t n, uint64_t out_p1[18]) {
    if (!m->n) return KZG_ERR_NO_SRS;
    const size_t k = m->kids.size();
    if (m->mode == kMultiReplicate) {  // the devices in turn: concurrent callers spread over the GPUs
        kzg_ctx* kid = m->kids[m->next_kid.fetch_add(1) % k];
        return is_mont ? kzg_commit(kid, (const uint64_t*)scalars, n, out_p1) : kzg_commit_le_bytes(kid, (const uint8_t*)scalars, n, out_p1);
    }
    std::lock_guard<std::mutex> lk(m->op_mu);
    CallTrace trace(m);
    m->last_error.clear();
    // coefficients beyond the SRS make the degree too high only when one of them is non-zero
    // (src/polynomial.rs:55-75, 201-205): same rule as the single-device context
    if (n > m->n) {
        if (fr_words_nonzero((const uint64_t*)scalars, m->n, n)) return KZG_ERR_DEGREE_TOO_HIGH;
        n = m->n;
    }
    std::vector<uint64_t> partials(18 * k, 0);
    int rc = for_each_kid(m, [&](size_t g) {
        const size_t lo = m->lo[g], hi = m->hi[g] < n ? m->hi[g] : n;
        if (hi <= lo) return (int)KZG_OK;  // nothing of the polynomial falls into this slice: partial = infinity
        const char* src = (const char*)scalars + lo * 32;
        return is_mont ? kzg_commit(m->kids[g], (const uint64_t*)src, hi - lo, partials.data() + 18 * g)
                       : kzg_commit_le_bytes(m->kids[g], (const uint8_t*)src, hi - lo, partials.data() + 18 * g);
    });
    if (rc != KZG_OK) return rc;
    return exchange_and_sum(m, partials.data(), 1, out_p1);
}

int multi_open(MultiState* m, const uint64_t* coeffs, size_t n, const uint64_t z[4], const uint64_t y[4], uint64_t out_p1[18]) {
    if (!m->n) return KZG_ERR_NO_SRS;
    const size_t k = m->kids.size();
    if (m->mode == kMultiReplicate) return kzg_open(m->kids[m->next_kid.fetch_add(1) % k], coeffs, n, z, y, out_p1);
    std::lock_guard<std::mutex> lk(m->op_mu);
    CallTrace trace(m);
    m->last_error.clear();
    std::memset(out_p1, 0, 144);
    hf::Fr zf, yf;
    std::memcpy(zf.l, z, 32);
    std::memcpy(yf.l, y, 32);
    if (n == 0) return yf.is_zero() ? KZG_OK : KZG_ERR_CONSTANT_POLY;  // [] - [y]  (src/polynomial.rs:138-143)
    // truncate trailing zeros (src/polynomial.rs:55-75): the quotient's length follows the last non-zero coefficient
    size_t n_eff = n;
    while (n_eff > 1 && !fr_words_nonzero(coeffs, n_eff - 1, n_eff)) n_eff--;
    if (n_eff == 1) {  // constant polynomial (src/polynomial.rs:159-167)
        return std::memcmp(coeffs, y, 32) == 0 ? KZG_OK : KZG_ERR_CONSTANT_POLY;
    }
    // quotient: n_eff - 1 coefficients q[j] paired with SRS[j]: the slices follow the SRS ranges; the last non-empty
    // slice also takes every coefficient above its range (c_{n_eff-1} legitimately sits one past the last SRS point;
    // anything beyond that makes the degree too high, decided after divide_by_root's own errors as in the reference)
    const size_t nq = n_eff - 1;
    std::vector<size_t> lo(k), hi(k);
    size_t last = 0;
    for (size_t g = 0; g < k; g++) {
        lo[g] = m->lo[g] < n_eff ? m->lo[g] : n_eff;
        hi[g] = m->hi[g] < n_eff ? m->hi[g] : n_eff;
        if (m->hi[g] > m->lo[g]) last = g;
    }
    hi[last] = n_eff;
    // 1. every device stages its slice and evaluates it: H_g = sum_{i in slice g} c_i z^(i - lo_g)
    std::vector<hf::Fr> H(k);
    std::vector<int> slot(k, -1);
    int rc = for_each_kid(m, [&](size_t g) {
        std::memset(H[g].l, 0, 32);
        if (hi[g] <= lo[g]) return (int)KZG_OK;
        return ctx_open_slice_begin(m->kids[g], coeffs + 4 * lo[g], hi[g] - lo[g], z, H[g].l, &slot[g]);
    });
    auto abort_all = [&] {
        for (size_t g = 0; g < k; g++)
            if (slot[g] >= 0) ctx_open_slice_abort(m->kids[g], slot[g]);
    };
    if (rc != KZG_OK) {
        abort_all();
        return rc;
    }
    // 2. carries, top slice first: C_g = S[hi_g], S[lo_g] = H_g + z^(len_g) C_g
    std::vector<hf::Fr> carry(k), start(k);
    hf::Fr c = {{0, 0, 0, 0}};
    for (size_t g = k; g-- > 0;) {
        carry[g] = c;
        c = hf::fr_add(H[g], hf::fr_mul(hf::fr_pow(zf, hi[g] - lo[g]), c));
        start[g] = c;
    }
    if (!(start[0] == yf)) {  // P(z) != y (src/polynomial.rs:184-192)
        abort_all();
        return KZG_ERR_REMAINDER;
    }
    if (nq > m->n) {  // commit's error comes last (src/polynomial.rs:201-205)
        abort_all();
        return KZG_ERR_DEGREE_TOO_HIGH;
    }
    // 3. partial proofs: the staged slice extended by its carry, claimed value S[lo_g]
    //    (an all-zero extended slice with a zero carry is a "constant polynomial": it contributes infinity)
    std::vector<uint64_t> partials(18 * k, 0);
    rc = for_each_kid(m, [&](size_t g) {
        if (slot[g] < 0) return (int)KZG_OK;
        const int s = slot[g];
        slot[g] = -1;  // finish releases the slot whatever happens
        return ctx_open_slice_finish(m->kids[g], s, hi[g] - lo[g], carry[g].l, z, start[g].l, partials.data() + 18 * g);
    });
    if (rc != KZG_OK) return rc;
    return exchange_and_sum(m, partials.data(), 1, out_p1);
}

int multi_commit_batch(MultiState* m, const uint64_t* coeffs, size_t n, size_t batch, size_t stride, uint64_t* out_p1s) {
    if (!m->n) return KZG_ERR_NO_SRS;
    if (n > m->n) return KZG_ERR_DEGREE_TOO_HIGH;  // batches take truncated polynomials only
    if (!batch) return KZG_OK;
    const size_t k = m->kids.size();
    std::lock_guard<std::mutex> lk(m->op_mu);
    m->last_error.clear();
    if (m->mode == kMultiReplicate)  // polynomial p -> device p mod K, nothing to exchange
        return for_each_kid(m, [&](size_t g) {
            const size_t mine = g < batch ? (batch - g + k - 1) / k : 0;
            return ctx_commit_batch_host(m->kids[g], coeffs, n, stride, g, k, mine, out_p1s);
        });
    // range mode: every polynomial is sharded; device g commits slice g of ALL of them in batched passes, then one
    // exchange carries the batch's partial sums
    std::vector<uint64_t> partials(18 * k * batch, 0);
    int rc = for_each_kid(m, [&](size_t g) {
        const size_t lo = m->lo[g], hi = m->hi[g] < n ? m->hi[g] : n;
        if (hi <= lo) return (int)KZG_OK;
        return ctx_commit_batch_host(m->kids[g], coeffs + 4 * lo, hi - lo, stride, 0, 1, batch, partials.data() + 18 * g * batch);
    });
    if (rc != KZG_OK) return rc;
    return exchange_and_sum(m, partials.data(), batch, out_p1s);
}

int multi_open_batch(MultiState* m, const uint64_t* coeffs, size_t n, size_t batch, size_t stride, const uint64_t* zs,
                     const uint64_t* ys, uint64_t* out_p1s, int* statuses) {
    if (!m->n) return KZG_ERR_NO_SRS;
    if (n - 1 > m->n) return KZG_ERR_DEGREE_TOO_HIGH;
    if (!batch) return KZG_OK;
    const size_t k = m->kids.size();
    if (m->mode == kMultiReplicate) {
        std::lock_guard<std::mutex> lk(m->op_mu);
        m->last_error.clear();
        return for_each_kid(m, [&](size_t g) {
            const size_t mine = g < batch ? (batch - g + k - 1) / k : 0;
            return ctx_open_batch_host(m->kids[g], coeffs, n, stride, g, k, mine, zs, ys, out_p1s, statuses);
        });
    }
    // range mode: the carry recurrence is per polynomial; the openings go one after the other through the sharded path
    for (size_t p = 0; p < batch; p++) {
        const int rc = multi_open(m, coeffs + p * stride * 4, n, zs + 4 * p, ys + 4 * p, out_p1s + 18 * p);
        if (rc != KZG_OK && rc != KZG_ERR_CONSTANT_POLY && rc != KZG_ERR_REMAINDER) return rc;
        statuses[p] = rc;
    }
    return KZG_OK;
}

}  // namespace kzg
