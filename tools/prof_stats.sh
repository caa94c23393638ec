#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench workload (run through gpurun).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-stats}
shift
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $@"
rm -rf gpurun_out/prof_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- $B > gpurun_out/prof_$TAG.log 2>&1
tail -1 gpurun_out/prof_$TAG.log | cut -c1-600
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -r cat | cut -c1-200
