#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for c in 15 16 17; do
  KZG_MSM_C=$c timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 4 --degree 131072 --batch 8 --force-dist 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline()); c = l["config"]
print(json.dumps({"N": c["commitments_per_step"], "per_rank_value": round(l["value"], 1), "ms_per_step": round(l["ms_per_step"], 3), "accum_ms": round(l["roofline"]["avg_kernel_ms"], 3), "c": c["digit_bits"], "phase": {k: round(v, 2) for k, v in l["phase_ms"].items()}}))' || exit 1
done
for c in 16 17 18; do
  KZG_MSM_C=$c timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 4 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline()); c = l["config"]
print(json.dumps({"N": c["commitments_per_step"], "value": round(l["value"], 1), "ms_per_step": round(l["ms_per_step"], 3), "accum_ms": round(l["roofline"]["avg_kernel_ms"], 3), "c": c["digit_bits"], "proofs": round(l["opening_proofs_per_sec"], 1)}))' || exit 1
done
