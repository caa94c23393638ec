#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
out=gpurun_out/sweep4.jsonl
: > $out
run() {
  echo "== $*" | tee -a $out
  env "${@:2}" timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 24 --warmup 3 --slots $1 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
c = l["config"]
a = l["roofline"]["avg_kernel_ms"]; m = l["valu"]["mixed_additions_per_launch"]
print(json.dumps({"value": round(l["value"], 1), "accum_ms": round(a, 3), "ns_per_madd": round(a * 1e6 / m, 4),
  "recoding": c["recoding"], "c": c["digit_bits"], "ok": c["bit_exact_vs_golden"], "proofs": round(l["opening_proofs_per_sec"], 1),
  "phase": {k: round(v, 2) for k, v in l["phase_ms"].items()}}))' | tee -a $out
}
A=$PWD/tools/ab
for mode in windows naf; do
  run 1 KZG_MSM_RECODE=$mode
  run 1 KZG_MSM_RECODE=$mode KZG_ACCUM_LANES=131072 KZG_ACCUM_LDS_KB=0
  run 1 KZG_MSM_RECODE=$mode KZG_MI355X_LIB=$A/libkzg_d1.so
  run 1 KZG_MSM_RECODE=$mode KZG_MI355X_LIB=$A/libkzg_lb3.so
done
run 3 KZG_MSM_RECODE=windows
run 3 KZG_MSM_RECODE=windows KZG_ACCUM_LANES=131072 KZG_ACCUM_LDS_KB=0
run 3 KZG_MSM_RECODE=windows KZG_MI355X_LIB=$A/libkzg_lb3.so
