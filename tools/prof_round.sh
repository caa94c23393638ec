#!/bin/bash
# Full round profile: default bench line, rocprofv3 kernel stats of the same command, PMC traffic.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 python3 bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_default.log | cut -c1-400
rm -rf gpurun_out/prof_default
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_default.log 2>&1; echo "rocprof rc=$?"
find gpurun_out/prof_default -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-160
./tools/prof_pmc.sh > gpurun_out/prof_pmc.log 2>&1; tail -30 gpurun_out/prof_pmc.log
