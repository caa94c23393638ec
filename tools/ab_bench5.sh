#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
A=$PWD/tools/ab
for v in pf2 pf3; do
  KZG_MI355X_LIB=$A/libkzg_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/gpu_tests_$v.log 2>&1 || { tail -20 gpurun_out/gpu_tests_$v.log; exit 1; }
  echo "$v: $(tail -1 gpurun_out/gpu_tests_$v.log)"
done
run() {
  env "${@:2}" timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 --slots $1 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
print(sys.argv[1:], round(l["value"], 1), "accum_ms", round(l["roofline"]["avg_kernel_ms"], 3), "proofs", round(l["opening_proofs_per_sec"], 1))' "$@" || exit 1
}
for rep in 1 2; do
  for s in 3 1; do
    run $s X=base
    run $s KZG_MI355X_LIB=$A/libkzg_pf2.so
    run $s KZG_MI355X_LIB=$A/libkzg_pf3.so
  done
done
run 3 KZG_MI355X_LIB=$A/libkzg_pf3.so KZG_ACCUM_LANES=262144
run 1 KZG_MI355X_LIB=$A/libkzg_pf3.so KZG_ACCUM_LANES=262144
