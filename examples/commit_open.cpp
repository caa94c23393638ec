// commit_open.cpp -- C++ caller of the C-ABI through include/kzg_mi355x.hpp: the shape of the
// reference's run_kate_proof_test (src/lib.rs:16-33) minus the pairing check.
// Build:  g++ -std=c++17 -Iinclude examples/commit_open.cpp -Lkzg_poly_commit_exploration_amd -lkzg_mi355x -o examples/commit_open
// Run  :  LD_LIBRARY_PATH=kzg_poly_commit_exploration_amd ./examples/commit_open   (needs an MI355X)
// Prints hex(commitment) and hex(proof) for P(x) = 1 + x (Montgomery one twice), z = y-consistent.
#include <cstdio>

#include "kzg_mi355x.hpp"

int main() {
    using namespace kzg_api;
    try {
        SetupArtifacts setup(0);
        std::array<uint8_t, 32> secret{};
        for (int i = 0; i < 32; i++) secret[i] = (uint8_t)i;  // benches/polynomial_commitment.rs:17-20
        setup.generate(secret, 16);
        // R mod r = Montgomery form of 1 (blst_fr)
        Scalar one{{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};
        Polynomial p = Polynomial::try_from({one, one, Scalar{}});  // 1 + x, trailing zero dropped
        G1Point c = p.commit(setup);
        Evaluation ev{one, p.evaluate_at(one, setup)};  // z = 1, y = P(1) = 2
        G1Point proof = ev.generate_proof(p, setup);
        auto hex = [](const std::array<uint8_t, 48>& b) {
            for (auto v : b) std::printf("%02x", v);
            std::printf("\n");
        };
        std::printf("degree %u\n", p.degree());
        hex(c.compress());
        hex(proof.compress());
    } catch (const Error& e) {
        std::fprintf(stderr, "kzg error %d: %s\n", e.status, e.what());
        return 1;
    }
    return 0;
}
