#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for tag in cur prev; do
  if [ $tag = prev ]; then export KZG_MI355X_LIB=$PWD/tools/ab/libkzg_prev.so; else unset KZG_MI355X_LIB; fi
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 > gpurun_out/prof_$tag.log 2>&1 || exit 1
  echo "== $tag $(tail -1 gpurun_out/prof_$tag.log | python3 -c 'import sys,json; l=json.loads(sys.stdin.readline()); print(round(l["value"],1))')"
  f=$(ls -t $(find gpurun_out/prof_$tag -name "*kernel_stats.csv") | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0][-26:]
    if any(k in n for k in ("window_double", "normalize", "srs_points", "gtable")): continue
    print("%-26s calls %4s avg_us %8.1f total_ms %8.2f" % (n, r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
