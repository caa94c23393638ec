#!/usr/bin/env python3
"""Copies the summaries tools/prof_round3.sh left under gpurun_out/ into profiles/r03_* (tracked), adding the hash of
the kernel sources they were measured on (bench.py quotes the PMC figures only while that hash matches).
Run right after the gpurun call, before touching msm_accum.hip / g1_30.hip.h / field30.hip.h again."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash)

G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
h = bench.kernel_source_hash()

line = [l for l in open(os.path.join(G, "r03_bench_default.log")).read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
with open(os.path.join(P, "r03_bench.jsonl"), "w") as f:
    f.write(line + "\n")
shutil.copy(os.path.join(G, "r03_kernel_stats.csv"), os.path.join(P, "r03_kernel_stats.csv"))

pm = json.load(open(os.path.join(G, "pmc_summary.json")))
acc = next(v for k, v in pm.items() if "accumulate" in k)
cal = next((v for k, v in pm.items() if "gather96" in k), {})
madds = d["valu"]["mixed_additions_per_launch"]
# calibration (MI355X_MICROARCH.md, HBM section): the gather of 96-byte records requests 1.5 128-B lines per record;
# FETCH_SIZE (KiB) tallies each 128-B request at 64 B -> x2, exactly as profiles/r01_traffic.json did
read_b = int(acc["FETCH_SIZE"] * 1024 * 2)
write_b = int(acc["WRITE_SIZE"] * 1024)
traffic = {
    "_comment": "HBM traffic of the accumulation kernel per launch (one degree-2^20 commitment), rocprofv3 --pmc passes of "
                "tools/prof_pmc.sh (FETCH_SIZE, WRITE_SIZE, request counters in separate passes); FETCH_SIZE x 2 per the gfx950 "
                "correction, confirmed on tools/microbench gather (1.51 GB requested as 96-byte records = 1.5 lines each -> "
                "FETCH_SIZE reads %.3f GB)" % (cal.get("FETCH_SIZE", 0) * 1024 / 1e9),
    "kernel": next(k for k in pm if "accumulate" in k), "kernel_source_hash": h,
    "fetch_size_kb": acc["FETCH_SIZE"], "write_size_kb": acc["WRITE_SIZE"], "tcc_ea0_rdreq": acc.get("TCC_EA0_RDREQ_sum"),
    "hbm_read_bytes_corrected": read_b, "hbm_write_bytes": write_b, "hbm_bytes_per_launch": read_b + write_b,
    "algorithmic_bytes_per_launch": d["roofline"]["algorithmic_bytes_per_launch"], "batch": 1, "recoding": d["config"]["recoding"],
}
json.dump(traffic, open(os.path.join(P, "r03_traffic.json"), "w"), indent=1)

v = json.load(open(os.path.join(G, "valu_summary.json")))
v["kernel_source_hash"] = h
v["mixed_additions_per_launch"] = madds
if "SQ_INSTS_VALU" in v and madds:
    v["valu_instructions_per_mixed_addition"] = v["SQ_INSTS_VALU"] / (madds / 64.0)
v["_comment"] = "VALU counters of the accumulation kernel (tools/prof_valu.sh, --steps 4 --slots 1, same workload)"
json.dump(v, open(os.path.join(P, "r03_valu_pmc.json"), "w"), indent=1)
print("profiles/r03_* written for kernel sources", h)
print("kernel avg ms (bench events): %.3f   traffic %.2f GB   VALUBusy %s" % (
    d["roofline"]["avg_kernel_ms"], (read_b + write_b) / 1e9, v.get("VALUBusy")))
with open(os.path.join(P, "r03_kernel_stats.csv")) as f:
    for i, row in enumerate(csv.reader(f)):
        if i < 4:
            print(row[0][:60], row[1:4])
