#!/bin/bash
# kernel-trace stats of degree-N commitments run one at a time: per-kernel average durations
# usage: tools/prof_stats_commit.sh <tag> [degree] [op]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-commit}
deg=${2:-1048576}
op=${3:-commit}
rm -rf gpurun_out/st_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$tag -- python3 tools/prof_latency.py $deg $op 10 > gpurun_out/st_$tag.log 2>&1
f=$(ls gpurun_out/st_$tag/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-50s calls %4s avg_us %9.1f" % (r["Name"].split("(")[0][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
