#!/bin/bash
# PMC comparison of the accumulation kernel under the two scalar recodings (windows: 2 GB table, NAF: 34 GB table).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python3 bench.py --steps 4 --warmup 1 --slots 1 --no-cpu-baseline"
for mode in windows naf; do
  export KZG_MSM_RECODE=$mode
  i=0
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "VALUBusy" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); rm -rf gpurun_out/pm_${mode}_$i
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pm_${mode}_$i -- $B > gpurun_out/pm_${mode}_$i.log 2>&1 || echo "pass $mode $i ($grp) failed"
  done
done
python3 - <<'PY'
import csv, glob, json
for mode in ("windows", "naf"):
    res, dur = {}, []
    for f in glob.glob("gpurun_out/pm_%s_*/*/*counter_collection.csv" % mode):
        for r in csv.DictReader(open(f)):
            if "k_bucket_accumulate" in r["Kernel_Name"]:
                res.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {k: round(sum(v) / len(v), 1) for k, v in sorted(res.items())}
    out["kernel_us_profiled"] = round(sum(dur) / max(1, len(dur)), 1)
    print(mode, json.dumps(out))
PY
