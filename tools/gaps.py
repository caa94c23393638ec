#!/usr/bin/env python3
"""Summarises accumulate-kernel durations and the idle gaps between them from a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1]) for r in rows))
acc = [e for e in ev if e[2] == "k_bucket_accumulate"]
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(acc, acc[1:])]
durs = [(a[1] - a[0]) / 1e3 for a in acc]
print("n", len(acc))
print("acc durations us:", [round(d) for d in durs[8:48]])
print("gaps us:", [round(g) for g in gaps[8:47]])
tot = (acc[47][1] - acc[8][0]) / 1e3 / 39
print("avg period us over 39 commits: %.0f" % tot)
