#!/bin/bash
# kernel-trace stats of ten degree-2^20 openings run one at a time: per-kernel average durations without counters
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-open}
rm -rf gpurun_out/st_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$tag -- python3 tools/prof_latency.py 1048576 open 10 > gpurun_out/st_$tag.log 2>&1
f=$(ls gpurun_out/st_$tag/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %4s avg_us %9.1f" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
