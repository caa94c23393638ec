#!/bin/bash
# default bench with and without a set of environment knobs, alternating, on one box: tools/ab_env.sh KNOB=V [KNOB=V ...]
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2; do
  for side in base knobs; do
    if [ $side = base ]; then E=""; else E="$*"; fi
    env $E timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
print("%-40s commits %.1f proofs %.1f accum_ms %.3f" % (sys.argv[1], l["value"], l["opening_proofs_per_sec"], l["roofline"]["avg_kernel_ms"]))' "$side $E"
  done
done
