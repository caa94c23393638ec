#!/bin/bash
# Round-3 evidence, second part (after tools/prof_round3.sh): the small measurements that go under profiles/r03_*.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 200 python tests/perf_small_commit.py > gpurun_out/r03_small_commit.jsonl 2>&1
timeout -k 10 300 python tests/perf_host_pointer_path.py > gpurun_out/r03_host_pointer.jsonl 2>&1
timeout -k 10 600 python tools/multi_overhead.py 2> gpurun_out/multi_overhead.err | grep '^{' > gpurun_out/r03_multi_overhead.jsonl
timeout -k 10 120 ./tools/instr_rates > gpurun_out/r03_instr_rates.jsonl 2>&1
bash tools/prof_quotient.sh > gpurun_out/prof_quotient.log 2>&1
for d in 1048576 65536 16384; do timeout -k 10 120 python tools/prof_latency.py $d commit 30; timeout -k 10 120 python tools/prof_latency.py $d open 30; done > gpurun_out/r03_latency.jsonl 2>&1
tail -3 gpurun_out/r03_host_pointer.jsonl; tail -8 gpurun_out/r03_multi_overhead.jsonl; cat gpurun_out/r03_latency.jsonl
