#!/bin/bash
# Round-3 evidence in one go (run on an MI355X through gpurun; outputs under gpurun_out/, summaries are copied to
# profiles/r03_* afterwards by tools/collect_profiles.py):
#   1. default bench line                       2. rocprofv3 --kernel-trace --stats of the same command
#   3. HBM traffic of k_bucket_accumulate (FETCH_SIZE / WRITE_SIZE / request counters in separate --pmc passes,
#      calibrated on tools/microbench gather as MI355X_MICROARCH.md prescribes)
#   4. VALU counters of the same kernel
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > gpurun_out/r03_bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r03_bench_default.log | cut -c1-300
rm -rf gpurun_out/prof_default
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -- python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/prof_default.log 2>&1; echo "rocprof rc=$?"
find gpurun_out/prof_default -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/r03_kernel_stats.csv
head -8 gpurun_out/r03_kernel_stats.csv | cut -c1-160
./tools/prof_pmc.sh > gpurun_out/prof_pmc.log 2>&1; tail -25 gpurun_out/prof_pmc.log
./tools/prof_valu.sh > gpurun_out/prof_valu.log 2>&1; tail -25 gpurun_out/prof_valu.log
