#!/bin/bash
# parity subset under the scheduling knobs (each must stay bit-exact and must not hang)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
for envs in "KZG_ACCUM_LDS_KB=0" "KZG_SERIALIZE_ACCUM=0" "KZG_STREAM_PRIORITIES=0" "KZG_ACCUM_LANES=65536" "KZG_FINE_CHUNKS=64" "KZG_MSM_C=12" "KZG_SMALL_SORT=0" "KZG_MSM_RECODE=naf KZG_MSM_C=13" "KZG_SORT_PACKED=0" "KZG_SPREAD_STAGED=0" "KZG_SPREAD_STAGED=0 KZG_SORT_PACKED=0 KZG_MSM_C=16"; do
  echo "== $envs"
  env $envs timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bench_degrees or pipelined or batched or skewed or one_bucket or randomized" 2>&1 | tail -1 || exit 1
done
