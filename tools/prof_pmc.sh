#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, one counter group per pass
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -> separate passes; on gfx950
# FETCH_SIZE under-counts wide coalesced reads by 2x -> calibrate on the same access pattern).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python3 bench.py --steps 4 --warmup 1 --slots 1 --no-cpu-baseline --no-extras"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_$tag -- $B > gpurun_out/pmc_$tag.log 2>&1
  rm -rf gpurun_out/cal_$tag
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/cal_$tag -- ./tools/microbench gather > gpurun_out/cal_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
def load(d):
    out=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out
res={}
for tag in ("FETCH_SIZE","WRITE_SIZE","TCC_EA0_RDREQ_sum"):
    for kind in ("pmc","cal"):
        d=load("gpurun_out/%s_%s"%(kind,tag))
        for k,v in d.items():
            if "accumulate" in k or "gather96" in k:
                for c,x in v.items():
                    res.setdefault(k,{})[c]=sum(x)/len(x)
print(json.dumps(res, indent=1))
open("gpurun_out/pmc_summary.json","w").write(json.dumps(res, indent=1))
PY
cat gpurun_out/cal_FETCH_SIZE.log | grep gather96
