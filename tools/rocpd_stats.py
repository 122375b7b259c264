#!/usr/bin/env python3
"""Summarise a rocprofv3 run into the same table `--stats` prints (per-kernel calls / total / avg / min / max).

Accepts either the rocpd sqlite database rocprofv3 7.x writes by default (`*_results.db`) or a
`*_kernel_trace.csv` written with `--output-format csv`.  Usage:
    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/r01_bench_kernel_stats.csv
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    for name, start, end in cur.execute(f"select {name_col}, start, end from kernels"):
        yield name, end - start


def rows_from_csv(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            yield r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    acc = defaultdict(list)
    for name, dur in rows:
        acc[name].append(dur)
    total = sum(sum(v) for v in acc.values()) or 1
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, len(v), sum(v), f"{sum(v) / len(v):.1f}", f"{100.0 * sum(v) / total:.2f}", min(v), max(v)])


if __name__ == "__main__":
    main()
