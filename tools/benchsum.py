#!/usr/bin/env python3
"""Prints the key figures of bench.py JSON lines: tools/benchsum.py gpurun_out/a.log gpurun_out/b.log ..."""
import json
import sys

for fn in sys.argv[1:]:
    try:
        line = [l for l in open(fn).read().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:  # noqa: BLE001
        print("%-40s unreadable (%s)" % (fn, e))
        continue
    ph = d.get("phase_ms_queueing_inclusive", d.get("phase_ms", {}))
    print("%-40s %7.1f /s  %.3f ms/step  kernel %.3f ms  proofs %s /s  sort %.2f+%.2f reduce %.2f quotient %s" % (
        fn.split("/")[-1], d["value"], d["ms_per_step"], d.get("roofline", {}).get("avg_kernel_ms", 0),
        ("%.1f" % d["opening_proofs_per_sec"]) if d.get("opening_proofs_per_sec") else "-",
        ph.get("digits_ms", 0), ph.get("scatter_ms", 0), ph.get("reduce_ms", 0), d.get("quotient_ms")))
    if "host_pointer_commitments_per_sec" in d:
        print("%-40s host-pointer %.1f commits/s %.1f proofs/s   batch-of-8 openings %s /s" % (
            "", d["host_pointer_commitments_per_sec"], d["host_pointer_proofs_per_sec"],
            ("%.1f" % d["openings_batch8_per_sec"]) if d.get("openings_batch8_per_sec") else "-"))
