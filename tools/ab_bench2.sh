#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
A=$PWD/tools/ab
run() {
  env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
print(sys.argv[1:], round(l["value"], 1), "accum_ms", round(l["roofline"]["avg_kernel_ms"], 3), "proofs", round(l["opening_proofs_per_sec"], 1), {k: round(v, 2) for k, v in l["phase_ms"].items()})' "$@" || exit 1
}
for rep in 1 2; do
  run KZG_FINE_CHUNKS=2048
  run KZG_FINE_CHUNKS=1024
  run KZG_FINE_CHUNKS=256
  run KZG_FINE_CHUNKS=512
  run KZG_MI355X_LIB=$A/libkzg_prev.so
done
