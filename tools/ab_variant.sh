#!/bin/bash
# A/B of the shipped library against a variant build on the default bench, alternating runs on one box.
# usage: tools/ab_variant.sh <variant .so (path inside the repo)> [extra env for the variant, e.g. KZG_ACCUM_LANES=196608]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
V=$PWD/$1
shift
run() {
  env "${@:2}" timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 --slots $1 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
print(sys.argv[1:], round(l["value"], 1), "accum_ms", round(l["roofline"]["avg_kernel_ms"], 3), "proofs", round(l["opening_proofs_per_sec"], 1))' "$@" || exit 1
}
for rep in 1 2; do
  for s in 3 1; do
    run $s X=current
    run $s KZG_MI355X_LIB=$V "$@"
  done
done
