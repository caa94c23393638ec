cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 100 2500 16384; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lat_$d -- python3 tools/prof_latency.py $d commit 50 > gpurun_out/lat_$d.log 2>&1
f=$(find gpurun_out/lat_$d -name "*kernel_stats.csv" | head -1)
echo "== degree $d"; tail -2 gpurun_out/lat_$d.log; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.2f} pct {r["Percentage"]}')
PY
done
