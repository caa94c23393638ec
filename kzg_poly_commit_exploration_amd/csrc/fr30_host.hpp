// fr30_host.hpp -- the host's side of fr30.hip.h: a multiplier of the device scans, prepared once per opening.
#pragma once
#include "fr30.hip.h"
#include "host_fr.hpp"

namespace kzg {

// blst_fr image of w (w * 2^256 mod r, canonical) -> balanced digits of w * 2^270 mod r: the operand form under which
// fr30_mul(x * 2^256, .) = x * w * 2^256.  One host Montgomery product by the plain integer 2^270 mod r.
inline Fr30 fr30_arg_from_mont256(const kzg_host::Fr& v) {
    static const kzg_host::Fr k270 = {{0x00008d53ffff72acULL, 0x12708804e7de5d54ULL, 0x5508b00eb2ea2f21ULL, 0x10dc4aca9a522018ULL}};
    const kzg_host::Fr w = kzg_host::fr_mul(v, k270);  // v * 2^270 / 2^256
    uint32_t l[8];
    for (int i = 0; i < 4; i++) {
        l[2 * i] = (uint32_t)w.l[i];
        l[2 * i + 1] = (uint32_t)(w.l[i] >> 32);
    }
    return fr30_from_limbs(l);
}

}  // namespace kzg
