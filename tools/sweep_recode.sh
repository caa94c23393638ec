#!/bin/bash
# A/B of the scalar recodings and accumulation launch shapes at degree 2^20 (bench.py, no CPU baseline).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
out=gpurun_out/sweep_recode.jsonl
: > $out
run() {
  echo "== $*" | tee -a $out
  env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 4 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
c = l["config"]
print(json.dumps({"value": round(l["value"], 1), "ms": round(l["ms_per_step"], 3), "accum_ms": round(l["roofline"]["avg_kernel_ms"], 3),
  "recoding": c["recoding"], "c": c["digit_bits"], "buckets": c["buckets"], "madds": l["valu"]["mixed_additions_per_launch"],
  "proofs": round(l["opening_proofs_per_sec"], 1), "ok": c["bit_exact_vs_golden"], "phase": {k: round(v, 2) for k, v in l["phase_ms"].items()}}))' | tee -a $out
}
run KZG_MSM_RECODE=windows
run KZG_MSM_RECODE=naf
run KZG_MSM_RECODE=naf KZG_MSM_C=18
run KZG_MSM_RECODE=naf KZG_MSM_C=20
run KZG_MSM_RECODE=naf KZG_ACCUM_LANES=131072 KZG_ACCUM_LDS_KB=0
run KZG_MSM_RECODE=naf KZG_ACCUM_LANES=131072
run KZG_MSM_RECODE=windows KZG_ACCUM_LANES=131072 KZG_ACCUM_LDS_KB=0
