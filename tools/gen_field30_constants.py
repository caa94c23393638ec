#!/usr/bin/env python3
"""Generates the constants of csrc/field30.hip.h (signed radix-2^30 Fp): balanced digits of p, -p^-1 mod 2^30 and
the conversion constant 2^384 mod p (written to csrc/field30_c384.inc).  Run from the repo root."""
import os

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
B, N = 30, 13


def balanced(v, n=N):
    d = []
    for _ in range(n - 1):
        r = v & ((1 << B) - 1)
        if r >= 1 << (B - 1):
            r -= 1 << B
        d.append(r)
        v = (v - r) >> B
    d.append(v)
    return d


def fmt(ds):
    return ", ".join(("-0x%x" % -x) if x < 0 else ("0x%x" % x) for x in ds)


if __name__ == "__main__":
    print("PD  =", fmt(balanced(P)))
    print("N0  = 0x%x" % ((-pow(P, -1, 1 << B)) % (1 << B)))
    print("sum|PD| / 2^29 = %.3f" % (sum(abs(x) for x in balanced(P)) / 2.0 ** 29))
    c384 = pow(2, 384, P)
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "kzg_poly_commit_exploration_amd", "csrc", "field30_c384.inc")
    with open(out, "w") as f:
        f.write("// 2^384 mod p, balanced radix-2^30 digits (tools/gen_field30_constants.py)\n" + fmt(balanced(c384)) + "\n")
    print("wrote", os.path.normpath(out))
    def emit(name, comment, digits):
        path = os.path.join(here, "..", "kzg_poly_commit_exploration_amd", "csrc", name)
        with open(path, "w") as f:
            f.write("// " + comment + " (tools/gen_field30_constants.py)\n" + fmt(digits) + "\n")
        print("wrote", os.path.normpath(path))

    def centered(v):
        v %= P
        return v - P if v > P // 2 else v

    emit("field30_p_u30.inc", "p, unsigned radix-2^30 digits", [(P >> (B * i)) & ((1 << B) - 1) for i in range(N)])
    emit("field30_c1170.inc", "2^1170 mod p, balanced radix-2^30 digits: turns v^-1 into the Montgomery form of the inverse",
         balanced(centered(pow(2, 1170, P))))
    emit("field30_c780.inc", "2^780 mod p, balanced radix-2^30 digits: turns a plain integer into Montgomery form (R'^2)",
         balanced(centered(pow(2, 780, P))))
    emit("field30_half.inc", "(p - 1) / 2, balanced radix-2^30 digits (canonical form: the sign rule of the compressed encoding)",
         balanced((P - 1) // 2))
    one = pow(2, 390, P)
    if one > P // 2:
        one -= P
    out = os.path.join(here, "..", "kzg_poly_commit_exploration_amd", "csrc", "field30_one.inc")
    with open(out, "w") as f:
        f.write("// 2^390 mod p (the Montgomery one), balanced radix-2^30 digits (tools/gen_field30_constants.py)\n" + fmt(balanced(one)) + "\n")
    print("wrote", os.path.normpath(out))
