#!/bin/bash
# AddressSanitizer on the HOST side of the library (api.hip: argument handling, host field / G1 / pairing code);
# device code is compiled normally (GPU ASan is not available on this pool).  Runs the CPU-only tests against it.
set -e
cd "$(dirname "$0")/.."
make -C kzg_poly_commit_exploration_amd/csrc -j8 >/dev/null
OUT=${TMPDIR:-/tmp}/kzg_asan
mkdir -p "$OUT"
FLAGS="-DKZG_LAZY_FP -DKZG_FIPS_SQR -O1 -g -fPIC --offload-arch=gfx950 -std=c++17 -Wno-unused-result -Wno-unused-value"
hipcc $FLAGS -fsanitize=address -fno-gpu-sanitize -c kzg_poly_commit_exploration_amd/csrc/api.hip -o "$OUT/api.o"
B=kzg_poly_commit_exploration_amd/csrc/build
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -fsanitize=address -shared-libsan -o "$OUT/libkzg_asan.so" "$OUT/api.o" \
  $B/msm_sort.o $B/msm_accum.o $B/msm_finalize.o $B/msm_reduce.o $B/poly_kernels.o $B/srs_kernels.o $B/multi.o $B/srs_io.o \
  -L/opt/rocm/lib -lrccl
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$RT KZG_MI355X_LIB="$OUT/libkzg_asan.so" \
  python -m pytest tests/test_verify.py tests/test_abi.py -x -q
