#!/bin/bash
# the bench line and its kernel-trace statistics only (tools/prof_round3.sh without the PMC passes, for builds that did
# not touch the accumulation kernel: profiles/r03_traffic.json / r03_valu_pmc.json stay valid while its sources' hash does)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > gpurun_out/r03_bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r03_bench_default.log | cut -c1-300
rm -rf gpurun_out/prof_default
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/prof_default.log 2>&1; echo "rocprof rc=$?"
find gpurun_out/prof_default -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/r03_kernel_stats.csv
head -8 gpurun_out/r03_kernel_stats.csv | cut -c1-160
