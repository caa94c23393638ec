#!/usr/bin/env python3
"""Prints the gaps between consecutive k_bucket_accumulate dispatches and what ran inside them
(from a rocprofv3 --kernel-trace CSV).  Usage: timeline.py <kernel_trace.csv> [first_n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1]) for r in rows))
acc = [e for e in ev if e[2] == "k_bucket_accumulate"]
acc = acc[-lim:]
for a, b in zip(acc, acc[1:]):
    gap = (b[0] - a[1]) / 1e3
    inside = [(e[2], (e[0] - a[1]) / 1e3, (e[1] - e[0]) / 1e3) for e in ev if e[0] < b[0] and e[1] > a[1] and e[2] != "k_bucket_accumulate"]
    print("acc %.0f us | gap to next %.0f us | overlapping/in-gap kernels:" % ((a[1] - a[0]) / 1e3, gap))
    for name, rel, dur in inside:
        print("      %-20s starts %+8.0f us rel. to acc end, runs %6.0f us" % (name, rel, dur))
