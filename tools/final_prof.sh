cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/prof_round2.sh > gpurun_out/prof_round2.log 2>&1
timeout -k 10 120 ./tools/microbench quad > gpurun_out/r02_microbench_quad.jsonl 2>&1
timeout -k 10 200 python tests/perf_small_commit.py > gpurun_out/r02_small_commit.jsonl 2>&1
timeout -k 10 300 python tests/perf_host_pointer_path.py > gpurun_out/r02_host_pointer.jsonl 2>&1
echo done
