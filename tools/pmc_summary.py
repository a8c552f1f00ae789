#!/usr/bin/env python3
"""Summarise rocprofv3 output directories: per kernel, median of each PMC counter and of the dispatch duration.

    python tools/pmc_summary.py DIR [DIR ...] [--match SUBSTR]

Reads every *counter_collection.csv (one row per dispatch and counter) and *kernel_trace.csv under the directories.
Kernel names are shortened to the text before the first '('."""
import csv
import glob
import os
import statistics
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].replace("bge::", "")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    counters = defaultdict(lambda: defaultdict(list))
    durations = defaultdict(list)
    for d in args:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    counters[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    durations[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    names = sorted(set(counters) | set(durations))
    for k in names:
        if match and match not in k:
            continue
        parts = [f"{c}={statistics.median(v):.6g}" for c, v in sorted(counters[k].items())]
        dur = f"n={len(durations[k])} median_us={statistics.median(durations[k]):.2f}" if durations[k] else ""
        print(f"{k}: {dur} " + " ".join(parts))


if __name__ == "__main__":
    main()
