#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for mode in windows naf; do
  rm -rf gpurun_out/prof_$mode
  KZG_MSM_RECODE=$mode timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$mode -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --slots 1 > gpurun_out/prof_$mode.log 2>&1
  echo "== $mode rc=$?"
  f=$(ls -t $(find gpurun_out/prof_$mode -name "*kernel_stats.csv") | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-28s calls %4s avg_us %9.1f min_us %9.1f max_us %9.1f" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
