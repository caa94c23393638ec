// Host build of csrc/fr30.hip.h for tests/test_fr30.py (plain g++; the header is __host__ __device__ code).
// Test infrastructure only.
#include <stdint.h>
#include <string.h>

#include "../../kzg_poly_commit_exploration_amd/csrc/fr30.hip.h"
#include "../../kzg_poly_commit_exploration_amd/csrc/host_fr.hpp"
#include "../../kzg_poly_commit_exploration_amd/csrc/fr30_host.hpp"

using namespace kzg;

extern "C" {

void r30_mul(const int32_t* a, const int32_t* b, int32_t* r) {
    Fr30 x, y;
    memcpy(x.d, a, sizeof x.d);
    memcpy(y.d, b, sizeof y.d);
    Fr30 z = fr30_mul(x, y);
    memcpy(r, z.d, sizeof z.d);
}
void r30_norm(const int32_t* a, int32_t* r) {
    Fr30 x;
    memcpy(x.d, a, sizeof x.d);
    Fr30 z = fr30_norm(x);
    memcpy(r, z.d, sizeof z.d);
}
void r30_from_limbs(const uint32_t* l, int32_t* r) {
    Fr30 z = fr30_from_limbs(l);
    memcpy(r, z.d, sizeof z.d);
}
void r30_from_limbs_raw(const uint32_t* l, int32_t* r) {
    Fr30 z = fr30_from_limbs_raw(l);
    memcpy(r, z.d, sizeof z.d);
}
int r30_abs_to_limbs(const int32_t* a, uint32_t* l) {
    Fr30 x;
    memcpy(x.d, a, sizeof x.d);
    return fr30_abs_to_limbs(x, l) ? 1 : 0;
}
void r30_to_limbs(const int32_t* a, uint32_t* l) {
    Fr30 x;
    memcpy(x.d, a, sizeof x.d);
    fr30_to_limbs(x, l);
}
// the host's preparation of a multiplier: blst_fr image (x * 2^256) -> digits of x * 2^270
void r30_arg_from_mont256(const uint64_t* limbs, int32_t* r) {
    kzg_host::Fr v;
    memcpy(v.l, limbs, 32);
    Fr30 z = fr30_arg_from_mont256(v);
    memcpy(r, z.d, sizeof z.d);
}
// largest |column| / 2^48 of fr30_mul in exact arithmetic (sum of magnitudes: the worst sign pattern)
int64_t r30_mul_max_column(const int32_t* a, const int32_t* b) {
    __int128 worst = 0;
    for (int k = 0; k < 17; k++) {
        __int128 mag = (__int128)1 << 34;  // carry in
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j < 0 || j > 8) continue;
            __int128 t = (__int128)a[i] * b[j];
            mag += t < 0 ? -t : t;
            int32_t rd = fr30_rd(j);
            mag += ((__int128)1 << 29) * (rd < 0 ? -(__int128)rd : rd);  // |m_i| <= 2^29
        }
        if (mag > worst) worst = mag;
    }
    return (int64_t)(worst >> 48);
}
}
