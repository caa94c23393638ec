for c in 17 18 19 16 17; do
 KZG_MSM_C=$c timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys, json
l = json.loads(sys.stdin.readline())
print("c", sys.argv[1], round(l["value"], 1), "accum_ms", round(l["roofline"]["avg_kernel_ms"], 3), "ms/step", round(l["ms_per_step"],3), l.get("phase_ms_queueing_inclusive"))' $c
done
