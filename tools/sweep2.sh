#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
out=gpurun_out/sweep2.jsonl
: > $out
run() {
  echo "== $*" | tee -a $out
  env "${@:2}" timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --slots $1 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
c = l["config"]
print(json.dumps({"value": round(l["value"], 1), "ms": round(l["ms_per_step"], 3), "accum_ms": round(l["roofline"]["avg_kernel_ms"], 3),
  "recoding": c["recoding"], "c": c["digit_bits"], "buckets": c["buckets"], "madds": l["valu"]["mixed_additions_per_launch"],
  "phase": {k: round(v, 2) for k, v in l["phase_ms"].items()}}))' | tee -a $out
}
run 1 KZG_MSM_RECODE=windows
run 1 KZG_MSM_RECODE=naf
run 1 KZG_MSM_RECODE=naf KZG_MSM_C=17
run 1 KZG_MSM_RECODE=naf KZG_MSM_C=20
