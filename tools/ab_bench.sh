#!/bin/bash
# A/B of the current build against tools/ab/libkzg_prev.so on the default bench (alternating runs on one box).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
A=$PWD/tools/ab
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/gpu_tests_ab.log 2>&1 || { tail -20 gpurun_out/gpu_tests_ab.log; exit 1; }
tail -1 gpurun_out/gpu_tests_ab.log
run() {
  env "${@:2}" timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 --slots $1 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
print(sys.argv[1:], round(l["value"], 1), "accum_ms", round(l["roofline"]["avg_kernel_ms"], 3), "proofs", round(l["opening_proofs_per_sec"], 1))' "$@" || exit 1
}
for rep in 1 2; do
  for s in 3 1; do
    run $s X=current
    run $s KZG_MI355X_LIB=$A/libkzg_prev.so
  done
done
