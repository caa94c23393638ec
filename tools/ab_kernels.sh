#!/bin/bash
# accumulation-kernel duration (rocprofv3 kernel trace) of variant builds: tools/ab_kernels.sh <lib.so>[:ENV=V[,ENV=V]] ...
cd "${GRAFT_REPO_ROOT:-.}"
for spec in "$@"; do
  lib=${spec%%:*}; envs=""
  [ "$spec" != "$lib" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  echo "== $spec"
  env KZG_MI355X_LIB=$PWD/$lib $envs bash tools/prof_stats_commit.sh v 2>&1 | grep -E "accumulate" || exit 1
done
