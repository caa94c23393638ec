// multi.hip -- a context that spans several devices of one node (kzg_ctx_create_multi, include/kzg_mi355x.h).
//
// The reference passes the whole SRS by slice to Polynomial::commit / Evaluation::generate_proof
// (src/polynomial.rs:200-215, 260-269).  Here device g keeps SRS points [lo_g, hi_g) resident with their window
// tables ("commitments shard naturally by SRS point range") and every call on the context is
// sharded transparently:
//   commit  sum_i c_i SRS_i = sum_g sum_{i in slice g} c_i SRS_i: one partial MSM per device, in parallel host
//           threads (one per device, each driving its own streams), then the exchange below;
//   open    q[j] = S[j+1] with S[i] = sum_{k>=i} c_k z^(k-i) (src/polynomial.rs:168-179): device g evaluates its slice
//           (the quotient scan without output), the host runs the K-step recurrence C_{g-1} = H_g + z^(len_g) C_g
//           (32 bytes per device), device g opens its slice extended by the carry C_g as one more top coefficient;
//           the partial proofs combine like partial commitments.  No kernel knows about the sharding.
//   exchange  "RCCL reduce of partial sums": RCCL has no user-defined reduction for curve points, so reduce =
//           ncclAllGather of the 144-byte blst_p1 partials (one per device, librccl linked directly, single
//           process: ncclCommInitAll + one group call) + K-1 complete additions (kzg_g1_sum).  When a device appears
//           more than once in the list (virtual slices on one GPU: how a single-GPU box rehearses the path) no
//           communicator can be formed and the partials are gathered on the host.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kzg_mi355x.h"
#include "engine.h"
#include "host_fr.hpp"

namespace hf = kzg_host;

namespace kzg {

struct MultiState {
    std::vector<kzg_ctx*> kids;
    std::vector<int> devices;
    std::vector<size_t> lo, hi;  // SRS range of every kid
    size_t n = 0;
    // exchange
    bool rccl = false;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<uint64_t*> d_in, d_out;  // 18 and 18 * K words per device
    uint64_t* h_gather = nullptr;        // pinned, 18 * K words
    std::string last_error;
    uint64_t rccl_exchanges = 0;
};

static void shard_range(size_t n, size_t g, size_t k, size_t& lo, size_t& hi) {
    const size_t per = (n + k - 1) / k;
    lo = g * per < n ? g * per : n;
    hi = lo + per < n ? lo + per : n;
}

int multi_create(const int* devices, int ndev, MultiState** out, std::string& err) {
    *out = nullptr;
    if (!devices || ndev <= 0) return KZG_ERR_INVALID_ARG;
    MultiState* m = new MultiState();
    for (int g = 0; g < ndev; g++) {
        kzg_ctx* kid = nullptr;
        int rc = kzg_ctx_create(devices[g], &kid);
        if (rc != KZG_OK) {
            for (auto* k : m->kids) kzg_ctx_destroy(k);
            delete m;
            return rc;
        }
        m->kids.push_back(kid);
        m->devices.push_back(devices[g]);
    }
    m->lo.assign(ndev, 0);
    m->hi.assign(ndev, 0);
    // a communicator needs distinct devices; one device needs none
    // (KZG_MULTI_FORCE_RCCL=1 forms the communicator for a single device too: a world of one, which is how the GPU
    // suite exercises the exchange code on a one-GPU box)
    std::set<int> distinct(devices, devices + ndev);
    const char* force = std::getenv("KZG_MULTI_FORCE_RCCL");
    if ((ndev > 1 || (force && force[0] == '1')) && (int)distinct.size() == ndev) {
        m->comms.resize(ndev);
        ncclResult_t r = ncclCommInitAll(m->comms.data(), ndev, devices);
        if (r != ncclSuccess) {
            err = std::string("ncclCommInitAll: ") + ncclGetErrorString(r);
            m->comms.clear();
            for (auto* k : m->kids) kzg_ctx_destroy(k);
            delete m;
            return KZG_ERR_HIP;
        }
        m->rccl = true;
        m->streams.resize(ndev);
        m->d_in.resize(ndev);
        m->d_out.resize(ndev);
        bool ok = true;
        for (int g = 0; g < ndev && ok; g++) {
            ok = hipSetDevice(devices[g]) == hipSuccess &&
                 hipStreamCreateWithFlags(&m->streams[g], hipStreamNonBlocking) == hipSuccess &&
                 hipMalloc((void**)&m->d_in[g], 18 * 8) == hipSuccess &&
                 hipMalloc((void**)&m->d_out[g], (size_t)ndev * 18 * 8) == hipSuccess;
        }
        ok = ok && hipHostMalloc((void**)&m->h_gather, (size_t)ndev * 18 * 8) == hipSuccess;
        if (!ok) {
            err = "multi-device exchange buffers";
            (void)hipGetLastError();
            // fall through to destroy: releases whatever was created
            extern void multi_destroy(MultiState*);
            multi_destroy(m);
            return KZG_ERR_HIP;
        }
    }
    *out = m;
    return KZG_OK;
}

void multi_destroy(MultiState* m) {
    if (!m) return;
    for (size_t g = 0; g < m->devices.size(); g++) {
        (void)hipSetDevice(m->devices[g]);
        if (g < m->d_in.size() && m->d_in[g]) (void)hipFree(m->d_in[g]);
        if (g < m->d_out.size() && m->d_out[g]) (void)hipFree(m->d_out[g]);
        if (g < m->streams.size() && m->streams[g]) (void)hipStreamDestroy(m->streams[g]);
        if (g < m->comms.size() && m->comms[g]) (void)ncclCommDestroy(m->comms[g]);
    }
    if (m->h_gather) (void)hipHostFree(m->h_gather);
    for (auto* k : m->kids) kzg_ctx_destroy(k);
    delete m;
}

size_t multi_srs_len(const MultiState* m) { return m->n; }
int multi_num_devices(const MultiState* m) { return (int)m->kids.size(); }
uint64_t multi_rccl_exchanges(const MultiState* m) { return m->rccl_exchanges; }
kzg_ctx* multi_kid(MultiState* m, int g) { return m->kids[g]; }
const char* multi_last_error(const MultiState* m) { return m->last_error.c_str(); }

// fn(g) on every device in its own host thread; returns the first non-OK status in device order
template <class F>
static int for_each_kid(MultiState* m, F&& fn) {
    const size_t k = m->kids.size();
    std::vector<int> rc(k, KZG_OK);
    if (k == 1) {
        rc[0] = fn(0);
    } else {
        std::vector<std::thread> pool;
        for (size_t g = 1; g < k; g++) pool.emplace_back([&, g] { rc[g] = fn(g); });
        rc[0] = fn(0);
        for (auto& t : pool) t.join();
    }
    for (size_t g = 0; g < k; g++)
        if (rc[g] != KZG_OK) {
            m->last_error = std::string("device slice ") + std::to_string(g) + ": " + kzg_last_error(m->kids[g]);
            return rc[g];
        }
    return KZG_OK;
}

int multi_srs_generate(MultiState* m, const uint8_t secret_be[32], uint64_t first, size_t n) {
    const size_t k = m->kids.size();
    m->n = 0;
    for (size_t g = 0; g < k; g++) shard_range(n, g, k, m->lo[g], m->hi[g]);
    int rc = for_each_kid(m, [&](size_t g) {
        if (m->hi[g] == m->lo[g]) return (int)KZG_OK;  // empty slice: the kid simply holds no SRS
        return kzg_srs_generate_g1(m->kids[g], secret_be, first + m->lo[g], m->hi[g] - m->lo[g]);
    });
    if (rc == KZG_OK) m->n = n;
    return rc;
}

int multi_srs_load(MultiState* m, const void* first_g1, size_t stride, size_t n) {
    const size_t k = m->kids.size();
    m->n = 0;
    for (size_t g = 0; g < k; g++) shard_range(n, g, k, m->lo[g], m->hi[g]);
    int rc = for_each_kid(m, [&](size_t g) {
        if (m->hi[g] == m->lo[g]) return (int)KZG_OK;
        return kzg_srs_load_g1(m->kids[g], (const char*)first_g1 + m->lo[g] * stride, stride, m->hi[g] - m->lo[g]);
    });
    if (rc == KZG_OK) m->n = n;
    return rc;
}

int multi_srs_load_affine(MultiState* m, const void* affine_xy, size_t n) {
    const size_t k = m->kids.size();
    m->n = 0;
    for (size_t g = 0; g < k; g++) shard_range(n, g, k, m->lo[g], m->hi[g]);
    int rc = for_each_kid(m, [&](size_t g) {
        if (m->hi[g] == m->lo[g]) return (int)KZG_OK;
        return kzg_srs_load_affine(m->kids[g], (const char*)affine_xy + m->lo[g] * 96, m->hi[g] - m->lo[g]);
    });
    if (rc == KZG_OK) m->n = n;
    return rc;
}

int multi_srs_load_compressed(MultiState* m, const uint8_t* compressed, size_t n, size_t* bad_index) {
    const size_t k = m->kids.size();
    m->n = 0;
    for (size_t g = 0; g < k; g++) shard_range(n, g, k, m->lo[g], m->hi[g]);
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

// partials[g] (18 words each) -> their sum.  Over RCCL when the context owns a communicator.
static int exchange_and_sum(MultiState* m, const std::vector<uint64_t>& partials, uint64_t out_p1[18]) {
    const size_t k = m->kids.size();
    if (!m->rccl) return kzg_g1_sum(partials.data(), k, out_p1);
    for (size_t g = 0; g < k; g++) {
        if (hipSetDevice(m->devices[g]) != hipSuccess ||
            hipMemcpyAsync(m->d_in[g], partials.data() + 18 * g, 18 * 8, hipMemcpyHostToDevice, m->streams[g]) != hipSuccess) {
            m->last_error = "exchange: upload of a partial sum failed";
            return KZG_ERR_HIP;
        }
    }
    ncclResult_t r = ncclGroupStart();
    for (size_t g = 0; g < k && r == ncclSuccess; g++)
        r = ncclAllGather(m->d_in[g], m->d_out[g], 18, ncclUint64, m->comms[g], m->streams[g]);
    ncclResult_t r2 = ncclGroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess) {
        m->last_error = std::string("ncclAllGather: ") + ncclGetErrorString(r != ncclSuccess ? r : r2);
        return KZG_ERR_HIP;
    }
    // every device now holds all K partials; the host takes device 0's copy (and waits for the others' streams so
    // that the buffers can be reused)
    if (hipSetDevice(m->devices[0]) != hipSuccess ||
        hipMemcpyAsync(m->h_gather, m->d_out[0], k * 18 * 8, hipMemcpyDeviceToHost, m->streams[0]) != hipSuccess) {
        m->last_error = "exchange: download of the gathered partial sums failed";
        return KZG_ERR_HIP;
    }
    for (size_t g = 0; g < k; g++) {
        if (hipSetDevice(m->devices[g]) != hipSuccess || hipStreamSynchronize(m->streams[g]) != hipSuccess) {
            m->last_error = "exchange: stream synchronisation failed";
            return KZG_ERR_HIP;
        }
    }
    m->rccl_exchanges++;
    return kzg_g1_sum(m->h_gather, k, out_p1);
}

int multi_commit(MultiState* m, const void* scalars, int is_mont, size_t n, uint64_t out_p1[18]) {
    if (!m->n) return KZG_ERR_NO_SRS;
    const size_t k = m->kids.size();
    // coefficients beyond the SRS make the degree too high only when one of them is non-zero
    // (src/polynomial.rs:55-75, 201-205): same rule as the single-device context
    if (n > m->n) {
        const uint64_t* c = (const uint64_t*)scalars;
        for (size_t i = m->n; i < n; i++)
            if (c[4 * i] | c[4 * i + 1] | c[4 * i + 2] | c[4 * i + 3]) return KZG_ERR_DEGREE_TOO_HIGH;
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
    return exchange_and_sum(m, partials, out_p1);
}

int multi_open(MultiState* m, const uint64_t* coeffs, size_t n, const uint64_t z[4], const uint64_t y[4], uint64_t out_p1[18]) {
    if (!m->n) return KZG_ERR_NO_SRS;
    const size_t k = m->kids.size();
    std::memset(out_p1, 0, 144);
    hf::Fr zf, yf;
    std::memcpy(zf.l, z, 32);
    std::memcpy(yf.l, y, 32);
    if (n == 0) return yf.is_zero() ? KZG_OK : KZG_ERR_CONSTANT_POLY;  // [] - [y]  (src/polynomial.rs:138-143)
    // truncate trailing zeros (src/polynomial.rs:55-75): the quotient's length follows the last non-zero coefficient
    size_t n_eff = n;
    while (n_eff > 1 && !(coeffs[4 * (n_eff - 1)] | coeffs[4 * (n_eff - 1) + 1] | coeffs[4 * (n_eff - 1) + 2] | coeffs[4 * (n_eff - 1) + 3]))
        n_eff--;
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
    // 1. H_g = sum_{i in slice g} c_i z^(i - lo_g)
    std::vector<hf::Fr> H(k);
    int rc = for_each_kid(m, [&](size_t g) {
        std::memset(H[g].l, 0, 32);
        if (hi[g] <= lo[g]) return (int)KZG_OK;
        return kzg_evaluate(m->kids[g], coeffs + 4 * lo[g], hi[g] - lo[g], z, H[g].l);
    });
    if (rc != KZG_OK) return rc;
    // 2. carries, top slice first: C_g = S[hi_g], S[lo_g] = H_g + z^(len_g) C_g
    std::vector<hf::Fr> carry(k), start(k);
    hf::Fr c = {{0, 0, 0, 0}};
    for (size_t g = k; g-- > 0;) {
        carry[g] = c;
        c = hf::fr_add(H[g], hf::fr_mul(hf::fr_pow(zf, hi[g] - lo[g]), c));
        start[g] = c;
    }
    if (!(start[0] == yf)) return KZG_ERR_REMAINDER;  // P(z) != y (src/polynomial.rs:184-192)
    if (nq > m->n) return KZG_ERR_DEGREE_TOO_HIGH;   // commit's error comes last (src/polynomial.rs:201-205)
    // 3. partial proofs: slice g extended by its carry, claimed value S[lo_g]
    std::vector<uint64_t> partials(18 * k, 0);
    rc = for_each_kid(m, [&](size_t g) {
        if (hi[g] <= lo[g]) return (int)KZG_OK;
        const size_t len = hi[g] - lo[g];
        std::vector<uint64_t> ext(4 * (len + 1));
        std::memcpy(ext.data(), coeffs + 4 * lo[g], len * 32);
        std::memcpy(ext.data() + 4 * len, carry[g].l, 32);
        // an all-zero extended slice with a zero carry would be a "constant polynomial": it contributes infinity
        int r = kzg_open(m->kids[g], ext.data(), len + 1, z, start[g].l, partials.data() + 18 * g);
        return r;
    });
    if (rc != KZG_OK) return rc;
    return exchange_and_sum(m, partials, out_p1);
}

}  // namespace kzg
