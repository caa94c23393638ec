#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for c in 16 17 18 19 20; do
  for s in 1 3; do
  KZG_MSM_C=$c timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 24 --warmup 3 --slots $s 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline()); c = l["config"]
a = l["roofline"]["avg_kernel_ms"]; m = l["valu"]["mixed_additions_per_launch"]
print(json.dumps({"c": c["digit_bits"], "slots": c["stream_slots"], "value": round(l["value"], 1), "accum_ms": round(a, 3), "madds": m, "ns_per_madd": round(a * 1e6 / m, 4), "phase": {k: round(v, 2) for k, v in l["phase_ms"].items()}}))' || exit 1
  done
done
