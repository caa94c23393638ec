#!/usr/bin/env python3
"""Regenerates the hand-scheduled v_mad_u64_u32 / v_addc_co_u32 groups at the top of tools/legacy_field/field_fips.hip.h
(macc{1..6}_vv, macc{1..6}_vv_first, macc{1..6}_vs).

Schedule: N multiply-adds into one 64-bit accumulator pair, each writing its carry to one of three SGPR pairs
(s[72:73], s[74:75], s[76:77]) in rotation; the v_addc that folds carry k into the overflow word is placed so that
at least two instructions separate it from the v_mad that produced the carry (VALU write of an SGPR -> VALU read
needs two wait states; inline asm is opaque to the hazard recogniser) and before the pair is overwritten by
multiply-add k+3.  N = 1 and N = 2 need explicit s_nop padding.  "_first": the first v_addc WRITES the overflow word
(0 + 0 + carry).  "_vs": the second factors are wave-uniform (SGPR / literal) operands.

Usage: python3 tools/gen_fips_groups.py            prints the generated functions
       python3 tools/gen_fips_groups.py --check    verifies that field_fips.hip.h contains exactly these bodies
"""
import os
import re
import sys

PAIRS = ["s[72:73]", "s[74:75]", "s[76:77]"]
CLOBBER = '"vcc", "s72", "s73", "s74", "s75", "s76", "s77"'


def schedule(n):
    """Returns the instruction order as a list of ('mad', k) / ('addc', k) / ('nop', count)."""
    if n == 1:
        return [("mad", 0), ("nop", 1), ("addc", 0)]
    if n == 2:
        return [("mad", 0), ("mad", 1), ("nop", 0), ("addc", 0), ("addc", 1)]
    out = [("mad", 0), ("mad", 1), ("mad", 2)]
    for k in range(3, n):
        out += [("addc", k - 3), ("mad", k)]
    for k in range(max(0, n - 3), n):
        out.append(("addc", k))
    return out


def body(n, first):
    x0 = 2
    y0 = 2 + n
    lines = []
    wrote_ovf = False
    for kind, k in schedule(n):
        if kind == "mad":
            lines.append("v_mad_u64_u32 %%0, %s, %%%d, %%%d, %%0" % (PAIRS[k % 3], x0 + k, y0 + k))
        elif kind == "nop":
            lines.append("s_nop %d" % k)
        else:
            src = "0" if (first and not wrote_ovf) else "%1"
            wrote_ovf = True
            lines.append("v_addc_co_u32 %%1, vcc, 0, %s, %s" % (src, PAIRS[k % 3]))
    return "\\n\\t".join(lines)


def function(n, kind):
    first = kind == "vv_first"
    ycons = '"s"' if kind == "vs" else '"v"'
    name = "macc%d_%s" % (n, kind)
    xs = ", ".join("u32 x%d" % i for i in range(n))
    ys = ", ".join("u32 y%d" % i for i in range(n))
    ins = ", ".join('"v"(x%d)' % i for i in range(n)) + ", " + ", ".join('%s(y%d)' % (ycons, i) for i in range(n))
    ovf = '"=&v"(ovf)' if first else '"+v"(ovf)'
    return ('KZG_DEV void %s(u64& acc, u32& ovf, %s, %s) {\n    asm volatile("%s"\n                 : "+v"(acc), %s\n'
            '                 : %s\n                 : %s);\n}' % (name, xs, ys, body(n, first), ovf, ins, CLOBBER))


def all_functions():
    out = []
    for kind in ("vv", "vv_first", "vs"):
        for n in range(1, 7):
            out.append(function(n, kind))
    return out


def asm_strings(text):
    return re.findall(r'KZG_DEV void (macc\d_\w+)\(.*?asm volatile\("(.*?)"\s*\n', text, flags=re.S)


if __name__ == "__main__":
    if "--check" in sys.argv:
        here = os.path.dirname(os.path.abspath(__file__))
        hdr = open(os.path.join(here, "legacy_field", "field_fips.hip.h")).read()
        have = dict(asm_strings(hdr))
        want = dict(asm_strings("\n".join(all_functions())))
        bad = [k for k in want if have.get(k) != want[k]]
        print("checked %d groups, %d differ" % (len(want), len(bad)))
        for k in bad:
            print(" ", k, "\n   have:", have.get(k), "\n   want:", want[k])
        sys.exit(1 if bad else 0)
    print("\n".join(all_functions()))
