#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
out=gpurun_out/sweep3.jsonl
: > $out
run() {
  echo "== $*" | tee -a $out
  env "${@:3}" timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --slots $1 --degree $2 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
c = l["config"]
a = l["roofline"]["avg_kernel_ms"]; m = l["valu"]["mixed_additions_per_launch"]
print(json.dumps({"value": round(l["value"], 1), "accum_ms": round(a, 3), "ns_per_madd": round(a * 1e6 / m, 4),
  "recoding": c["recoding"], "c": c["digit_bits"], "buckets": c["buckets"], "madds": m, "table_gib": c["table_gib"],
  "phase": {k: round(v, 2) for k, v in l["phase_ms"].items()}}))' | tee -a $out
}
for d in 65536 262144 1048576; do
  run 1 $d KZG_MSM_RECODE=windows
  run 1 $d KZG_MSM_RECODE=naf
done
