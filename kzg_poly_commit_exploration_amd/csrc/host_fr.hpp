// host_fr.hpp -- scalar field Fr on the host (4 x u64, Montgomery R = 2^256: the blst_fr memory image that crosses
// the C-ABI).  Used by the multi-device context for the K-step carry recurrence of a range-sharded opening
// (multi.hip); everything O(n) stays on the devices.  Mirrors the operations the reference takes from blst through
// `Scalar` (src/scalar.rs:111-117, 192-218): add, sub, mul, pow.
#pragma once
#include <stdint.h>
#include <string.h>

namespace kzg_host {

struct Fr {
    uint64_t l[4];
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const Fr& o) const { return memcmp(l, o.l, sizeof l) == 0; }
};

static const Fr kFrMod = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}};
static const Fr kFrOne = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};
static const uint64_t kFrN0 = 0xfffffffeffffffffULL;

inline bool fr_geq(const Fr& a, const Fr& b) {
    for (int i = 3; i >= 0; --i)
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    return true;
}
inline Fr fr_raw_sub(const Fr& a, const Fr& b, uint64_t& borrow) {
    Fr r;
    borrow = 0;
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 d = (unsigned __int128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return r;
}
inline Fr fr_add(const Fr& a, const Fr& b) {
    Fr s;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (unsigned __int128)a.l[i] + b.l[i];
        s.l[i] = (uint64_t)c;
        c >>= 64;
    }
    uint64_t br;
    return (c || fr_geq(s, kFrMod)) ? fr_raw_sub(s, kFrMod, br) : s;  // 2r < 2^256: c is always 0
}
inline Fr fr_mul(const Fr& a, const Fr& b) {  // Montgomery product, CIOS
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (unsigned __int128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * kFrN0;
        c = ((unsigned __int128)m * kFrMod.l[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) {
            c += (unsigned __int128)m * kFrMod.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r = {{t[0], t[1], t[2], t[3]}};
    uint64_t br;
    if (t[4] || fr_geq(r, kFrMod)) r = fr_raw_sub(r, kFrMod, br);
    return r;
}
inline Fr fr_pow(Fr base, uint64_t e) {
    Fr acc = kFrOne;
    while (e) {
        if (e & 1) acc = fr_mul(acc, base);
        base = fr_mul(base, base);
        e >>= 1;
    }
    return acc;
}

}  // namespace kzg_host
