#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun).  Writes under gpurun_out/.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python3 bench.py --steps 5 --warmup 2 --slots 1 --no-cpu-baseline"
echo "== inline"; timeout -k 10 300 $B > gpurun_out/bench_inline.log 2>&1; tail -1 gpurun_out/bench_inline.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phase_ms'])"
echo "== call"; KZG_ACCUM_VARIANT=call timeout -k 10 300 $B > gpurun_out/bench_call.log 2>&1; tail -1 gpurun_out/bench_call.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phase_ms'])"
echo "== rocprof stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $B > gpurun_out/prof_stats.log 2>&1
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -r head -30
echo "== rocprof pmc icache"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/prof_pmc1 -- $B > gpurun_out/prof_pmc1.log 2>&1
echo "== rocprof pmc sq"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_IFETCH SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_pmc2 -- $B > gpurun_out/prof_pmc2.log 2>&1
ls -R gpurun_out | head -50
