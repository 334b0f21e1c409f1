#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (avg per dispatch).
usage: pmc_summary.py out.txt dir1 [dir2 ...]"""
import collections, csv, glob, re, sys


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]

out = open(sys.argv[1], "w")
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        out.write("# %s\n" % f)
        for (k, c), v in sorted(agg.items()):
            out.write("%-62s %-24s dispatches=%d avg=%.6g\n" % (k, c, len(v), sum(v) / len(v)))
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        out.write("# %s (kernel durations, ms)\n" % f)
        for k, v in sorted(agg.items()):
            out.write("%-62s calls=%d avg_ms=%.4f min_ms=%.4f\n" % (k, len(v), sum(v) / len(v), min(v)))
out.close()
