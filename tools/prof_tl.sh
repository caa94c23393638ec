#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_tl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --no-cpu-baseline --no-extras --steps 16 --warmup 3 > gpurun_out/prof_tl.log 2>&1 || exit 1
f=$(ls -t $(find gpurun_out/prof_tl -name "*kernel_trace.csv") | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", ""), r.get("Queue_Id", "?")) for r in rows)
acc = [e for e in ev if e[2] == "k_bucket_accumulate"]
a0 = acc[10]
t0 = a0[0]
print("window: 3 accumulations starting at the 11th; times in us relative to its start; columns: start end dur queue name")
for e in ev:
    if e[0] >= t0 - 200000 and e[0] <= t0 + 9_000_000 and (e[1] - e[0] > 20000 or "heavy" in e[2] or "finalize" in e[2]):
        print("%8.0f %8.0f %7.0f  q%-3s %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[3], e[2]))
PY
