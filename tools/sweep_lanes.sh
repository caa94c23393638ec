#!/bin/bash
# accumulation grid shape: lanes (= segments) per launch, standalone (1 slot) and pipelined (all slots)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for lanes in ${LANES:-98304 131072 163840 196608 262144}; do
  for s in 1 0; do
    KZG_ACCUM_LANES=$lanes timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 --slots $s 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
print(json.dumps({"lanes": int(sys.argv[1]), "slots": int(sys.argv[2]), "value": round(l["value"], 1), "proofs": round(l.get("opening_proofs_per_sec", 0), 1), "accum_ms": round(l["roofline"]["avg_kernel_ms"], 3)}))' $lanes $s || exit 1
  done
done
