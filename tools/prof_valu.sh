#!/bin/bash
# VALU utilisation of the dominant kernel from PMC counters (separate passes; no trace domains mixed in).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python3 bench.py --steps 4 --warmup 1 --slots 1 --no-cpu-baseline --no-extras"
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT" "VALUBusy" "VALUUtilization"; do
  i=$((i+1)); rm -rf gpurun_out/valu_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/valu_$i -- $B > gpurun_out/valu_$i.log 2>&1 || echo "pass $i ($grp) failed"
done
python3 - <<'PY'
import csv, glob, collections, json
res = {}
dur = []
for f in glob.glob("gpurun_out/valu_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_bucket_accumulate" in r["Kernel_Name"]:
            res.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {k: sum(v) / len(v) for k, v in res.items()}
out["kernel_us_profiled"] = sum(dur) / max(1, len(dur))
simds = 256 * 4
if "SQ_ACTIVE_INST_VALU" in out and "GRBM_GUI_ACTIVE" in out:
    # gfx94x formula: VALUBusy = 100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / GRBM_GUI_ACTIVE (GUI_ACTIVE summed over 8 XCDs)
    out["VALUBusy_formula_percent"] = 100.0 * out["SQ_ACTIVE_INST_VALU"] * 4 / simds / (out["GRBM_GUI_ACTIVE"] / 8)
print(json.dumps(out, indent=1))
open("gpurun_out/valu_summary.json", "w").write(json.dumps(out, indent=1))
PY
