#!/usr/bin/env python3
"""Instruction histogram of k_bucket_accumulate's main loop from the compiler's assembly, priced with the issue costs
tools/instr_rates measured on the MI355X (profiles/r03_instr_rates.jsonl, 2 waves per SIMD).
  hipcc -DKZG_LAZY_FP -DKZG_FIPS_SQR -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S msm_accum.hip -o accum.s
  python3 tools/isa_histogram.py accum.s profiles/r03_instr_rates.jsonl > profiles/r03_accum_isa_histogram.txt"""
import collections
import json
import re
import sys

asm = open(sys.argv[1]).read().splitlines()
rates = {}
for line in open(sys.argv[2]):
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        if j["waves_per_simd"] == 2:
            rates[j["instr"]] = j["cost_vs_v_add_u32"]
start = next(i for i, l in enumerate(asm) if re.match(r"^_ZN3kzg19k_bucket_accumulate.*:\s*; @", l))
end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
body = asm[start:end]
# the main loop: from the first loop header whose body holds the LDS-DMA gathers to the back edge behind them
heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
dma = [i for i, l in enumerate(body) if "global_load_lds_dwordx4" in l]
head = max(h for h in heads if h < dma[-1])
name = body[head].split(":")[0].strip().lstrip(".L")          # e.g. BB3_10
labels = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
inside = [i for i in labels if ("Header=%s " % name) in body[i] + " " or ("Parent Loop %s " % name) in body[i] + " "]
first = min([head] + inside)
after = [i for i in labels if i > max(inside) and i not in inside]
last = (after[0] if after else len(body)) - 1
loop = [l.split()[0] for l in body[first:last + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
hist = collections.Counter(loop)
valu = {k: v for k, v in hist.items() if k.startswith("v_")}
def cost(op):
    base = op.replace("_e32", "").replace("_e64", "")
    if base in rates:
        return rates[base]
    if base.startswith(("v_mad_i64", "v_mad_u64")):
        return rates["v_mad_i64_i32"]
    three_operand = ("v_bfe", "v_alignbit", "v_add3", "v_lshl_add", "v_ashrrev_i64", "v_lshlrev_b64", "v_mul_lo", "v_mul_hi", "v_mad",
                     "v_or3", "v_and_or", "v_cndmask_b32_e64", "v_sub_co", "v_subb_co", "v_add_co", "v_addc_co", "v_cmp")
    return rates["v_bfe_i32"] if base.startswith(three_operand) else 1.0
print("k_bucket_accumulate, main loop (static count, one iteration = one mixed addition per lane; the rare branches --")
print("exact zero tests, doubling, bucket flush -- are in it): %d instructions, %d of them VALU" % (len(loop), sum(valu.values())))
print("issue cost in units of one v_add_u32 at 2 waves/SIMD (profiles/r03_instr_rates.jsonl); PMC: 4462 VALU executed per iteration")
print()
print("%-24s %6s %6s %8s" % ("instruction", "count", "cost", "units"))
total = 0.0
rows = sorted(valu.items(), key=lambda kv: -kv[1] * cost(kv[0]))
for op, n in rows:
    u = n * cost(op)
    total += u
for op, n in rows[:24]:
    print("%-24s %6d %6.2f %8.0f  %4.1f %%" % (op, n, cost(op), n * cost(op), 100 * n * cost(op) / total))
mad = sum(n * cost(op) for op, n in valu.items() if op.startswith("v_mad_i64"))
print()
print("multiply-adds: %.0f of %.0f units = %.1f %% of the loop's issue time; everything else %.1f %%" % (mad, total, 100 * mad / total, 100 - 100 * mad / total))
print("scalar / memory / LDS instructions in the loop:", {k: v for k, v in hist.most_common() if not k.startswith("v_")})
