#!/bin/bash
# A/B of two builds of the library on the default bench (alternating runs on the same box).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
A=$PWD/tools/ab
for rep in 1 2 3; do
  for lib in "" $A/libkzg_prev.so; do
    KZG_MI355X_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys, json, os
l = json.loads(sys.stdin.readline())
print(os.environ.get("KZG_MI355X_LIB") or "current", round(l["value"], 1), "accum_ms", round(l["roofline"]["avg_kernel_ms"], 3), "proofs", round(l["opening_proofs_per_sec"], 1), {k: round(v, 2) for k, v in l["phase_ms"].items()})' || exit 1
  done
done
