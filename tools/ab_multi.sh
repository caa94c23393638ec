#!/bin/bash
# default bench on several builds of the library, alternating, on one box: tools/ab_multi.sh <lib.so> [<lib.so> ...]
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
  for lib in current "$@"; do
    if [ "$lib" = current ]; then unset KZG_MI355X_LIB; else export KZG_MI355X_LIB=$PWD/$lib; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
print("%-50s commits %.1f proofs %.1f accum_ms %.3f" % (sys.argv[1], l["value"], l["opening_proofs_per_sec"], l["roofline"]["avg_kernel_ms"]))' "$lib"
  done
done
