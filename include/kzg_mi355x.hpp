// kzg_mi355x.hpp -- header-only C++ mirror of the reference's commit / open API over the C-ABI
// (include/kzg_mi355x.h).  The reference is compiled code (Rust); with no Rust toolchain in the build
// image this is the host side a C++ caller uses.  Type and method names follow the reference:
//   Scalar                  src/scalar.rs:7-8      (blst_fr memory image, Montgomery)
//   G1Point                 src/curves.rs:10-17    (blst_p1 memory image)
//   SetupArtifacts          src/trusted_setup.rs   (the G1 half of Vec<SetupArtifact>, resident on the GPU)
//   Polynomial::commit      src/polynomial.rs:200-215
//   Evaluation::generate_proof  src/polynomial.rs:260-269
// Errors are thrown as kzg::Error carrying the reference's anyhow message (kzg_strerror).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "kzg_mi355x.h"

namespace kzg_api {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};
inline void check(int rc, const kzg_ctx* ctx = nullptr) {
    if (rc == KZG_OK) return;
    std::string m = kzg_strerror(rc);
    if (rc == KZG_ERR_HIP && ctx) m += std::string(": ") + kzg_last_error(ctx);
    throw Error(rc, m);
}

struct Scalar {  // blst_fr: 4 x u64 little-endian limbs, Montgomery form
    std::array<uint64_t, 4> l{};
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }  // src/scalar.rs:221-223
    bool operator==(const Scalar& o) const { return l == o.l; }
};

struct G1Point {  // blst_p1: {x, y, z} Jacobian, Montgomery; z == 0 <=> infinity
    std::array<uint64_t, 18> p1{};
    std::array<uint8_t, 48> compress() const {  // Serialize for G1Point, src/curves.rs:99-110
        std::array<uint8_t, 48> out{};
        check(kzg_g1_compress(p1.data(), out.data()));
        return out;
    }
    bool is_infinity() const {
        uint64_t o = 0;
        for (int i = 12; i < 18; i++) o |= p1[i];
        return o == 0;
    }
    G1Point add(const G1Point& b) const {  // src/curves.rs:79-85
        uint64_t both[36];
        std::memcpy(both, p1.data(), 144);
        std::memcpy(both + 18, b.p1.data(), 144);
        G1Point r;
        check(kzg_g1_sum(both, 2, r.p1.data()));
        return r;
    }
};

class SetupArtifacts {  // owns one engine context = one GPU with the SRS resident
   public:
    explicit SetupArtifacts(int device = 0) { check(kzg_ctx_create(device, &ctx_)); }
    // one context over several devices: SRS split by point range, commit / open sharded transparently
    // replicate_srs: every device keeps the whole SRS and batches are split by polynomial (BASELINE config 5)
    explicit SetupArtifacts(const std::vector<int>& devices, bool replicate_srs = false) {
        check(kzg_ctx_create_multi_ex(devices.data(), (int)devices.size(), replicate_srs ? KZG_MULTI_REPLICATE_SRS : 0u, &ctx_));
    }
    ~SetupArtifacts() { kzg_ctx_destroy(ctx_); }
    SetupArtifacts(const SetupArtifacts&) = delete;
    SetupArtifacts& operator=(const SetupArtifacts&) = delete;
    // SetupArtifactsGenerator::new(secret).take(n), src/trusted_setup.rs:20-28, 40-62 (G1 side)
    void generate(const std::array<uint8_t, 32>& secret_be, size_t n, uint64_t first = 0) {
        check(kzg_srs_generate_g1(ctx_, secret_be.data(), first, n), ctx_);
    }
    // from the reference's own memory: &srs[0].g1, stride = sizeof(SetupArtifact)
    void load(const void* first_g1, size_t stride_bytes, size_t n) { check(kzg_srs_load_g1(ctx_, first_g1, stride_bytes, n), ctx_); }
    // the 48-byte points of the CLI's setup.json (src/curves.rs:99-183), decompressed on the device
    void load_compressed(const uint8_t* points48, size_t n) { check(kzg_srs_load_compressed(ctx_, points48, n, nullptr), ctx_); }
    void save(const std::string& path) const { check(kzg_srs_save(ctx_, path.c_str()), ctx_); }   // binary affine cache
    void load_file(const std::string& path) { check(kzg_srs_load_file(ctx_, path.c_str()), ctx_); }
    size_t len() const { return kzg_srs_len(ctx_); }
    size_t set_max_batch(size_t b) { check(kzg_set_max_batch(ctx_, b), ctx_); return kzg_max_batch(ctx_); }
    // setup_artifacts[index].g2 = [s^index]G2 (src/trusted_setup.rs:64-72), computed on the host: what verify_proof reads at index 1
    static std::array<uint64_t, 36> g2_at(const std::array<uint8_t, 32>& secret_be, uint64_t index = 1) {
        std::array<uint64_t, 36> out{};
        check(kzg_srs_g2_at(secret_be.data(), index, out.data()));
        return out;
    }
    kzg_ctx* ctx() const { return ctx_; }

   private:
    kzg_ctx* ctx_ = nullptr;
};

class Polynomial {
   public:
    // TryFrom<Vec<Scalar>>, src/polynomial.rs:55-75: trailing zeros dropped, index 0 kept
    static Polynomial try_from(std::vector<Scalar> v) {
        size_t last = 0;
        for (size_t i = 0; i < v.size(); i++)
            if (!v[i].is_zero()) last = i;
        if (!v.empty()) v.resize(last + 1);
        Polynomial p;
        p.coefficients_ = std::move(v);
        return p;
    }
    uint32_t degree() const { return coefficients_.empty() ? 0 : (uint32_t)(coefficients_.size() - 1); }  // :93-98
    const std::vector<Scalar>& coefficients() const { return coefficients_; }
    G1Point commit(const SetupArtifacts& setup) const {  // :200-215
        G1Point out;
        check(kzg_commit(setup.ctx(), reinterpret_cast<const uint64_t*>(coefficients_.data()), coefficients_.size(),
                         out.p1.data()), setup.ctx());
        return out;
    }
    Scalar evaluate_at(const Scalar& x, const SetupArtifacts& setup) const {  // :112-123
        Scalar y;
        check(kzg_evaluate(setup.ctx(), reinterpret_cast<const uint64_t*>(coefficients_.data()), coefficients_.size(),
                           x.l.data(), y.l.data()), setup.ctx());
        return y;
    }

   private:
    std::vector<Scalar> coefficients_;
};

// The loop of the reference's callers over many polynomials (src/lib.rs:16-33 once per polynomial) as ONE call:
// `batch` polynomials of n coefficients each, polynomial p at coeffs[p * n ..]; on a multi-device context the
// polynomials (replicated SRS) or every polynomial's ranges (range-split SRS) spread over the GPUs.
inline std::vector<G1Point> commit_batch(const SetupArtifacts& setup, const std::vector<Scalar>& coeffs, size_t n) {
    const size_t batch = n ? coeffs.size() / n : 0;
    std::vector<G1Point> out(batch);
    check(kzg_commit_batch(setup.ctx(), reinterpret_cast<const uint64_t*>(coeffs.data()), n, batch, n,
                           reinterpret_cast<uint64_t*>(out.data())), setup.ctx());
    return out;
}
// statuses[p]: KZG_OK, KZG_ERR_CONSTANT_POLY or KZG_ERR_REMAINDER -- what generate_proof would have returned for p
inline std::vector<G1Point> open_batch(const SetupArtifacts& setup, const std::vector<Scalar>& coeffs, size_t n,
                                       const std::vector<Scalar>& points, const std::vector<Scalar>& results, std::vector<int>& statuses) {
    const size_t batch = points.size();
    std::vector<G1Point> out(batch);
    statuses.assign(batch, KZG_OK);
    check(kzg_open_batch(setup.ctx(), reinterpret_cast<const uint64_t*>(coeffs.data()), n, batch, n,
                         reinterpret_cast<const uint64_t*>(points.data()), reinterpret_cast<const uint64_t*>(results.data()),
                         reinterpret_cast<uint64_t*>(out.data()), statuses.data()), setup.ctx());
    return out;
}
static_assert(sizeof(G1Point) == 144 && sizeof(Scalar) == 32, "memory images of blst_p1 / blst_fr");

struct Evaluation {  // src/polynomial.rs:249-253
    Scalar point, result;
    G1Point generate_proof(const Polynomial& polynomial, const SetupArtifacts& setup) const {  // :260-269
        G1Point out;
        const auto& c = polynomial.coefficients();
        check(kzg_open(setup.ctx(), reinterpret_cast<const uint64_t*>(c.data()), c.size(), point.l.data(),
                       result.l.data(), out.p1.data()), setup.ctx());
        return out;
    }
    // :276-294; s_g2 = setup_artifacts[1].g2 as blst_p2 (36 x u64).  Host-side pairing check.
    bool verify_proof(const G1Point& proof, const G1Point& commitment, const uint64_t s_g2[36]) const {
        int valid = 0;
        check(kzg_verify_proof(commitment.p1.data(), proof.p1.data(), point.l.data(), result.l.data(), s_g2, &valid), nullptr);
        return valid == 1;
    }
};

}  // namespace kzg_api
