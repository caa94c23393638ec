#!/bin/bash
# Per-rank shapes of the N-GPU bench on ONE GPU: degree 2^20 / N terms per rank, N commitments per step, live
# RCCL exchange at world size 1 (--force-dist); then the real multi-process path over gloo with 2 ranks sharing
# the device.  value x N is what N such GPUs would deliver if the exchange scaled.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
out=gpurun_out/emulate_scaling.jsonl
: > $out
for N in 1 2 4 8; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 30 --warmup 4 --degree $((1048576 / N)) --batch $N --force-dist 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline()); c = l["config"]
print(json.dumps({"N": c["commitments_per_step"], "terms_per_gpu": c["terms_per_gpu"], "per_rank_value": round(l["value"], 1), "ms_per_step": round(l["ms_per_step"], 3), "accum_ms": round(l["roofline"]["avg_kernel_ms"], 3), "c": c["digit_bits"]}))' | tee -a $out || exit 1
done
MASTER_ADDR=127.0.0.1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --backend gloo --device 0 2>&1 | tail -1 | cut -c1-400 | tee -a $out
